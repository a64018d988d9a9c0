// ffb6d_amd/csrc/errors.hip -- thread-local error text + ABI version for the C boundary.
#include "common.h"

namespace ffb6d {

char* err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ffb6d

extern "C" {

const char* ffb6d_last_error(void) { return ffb6d::err_buf(); }

int ffb6d_abi_version(void) { return 1000; }

}  // extern "C"
