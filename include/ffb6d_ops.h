/*
 * include/ffb6d_ops.h -- C ABI of the MI355X-native RandLA-Net neighbour ops and the
 * pixel<->point fusion gathers of FFB6D.  Every function is the device-side body of one
 * reference Python operator (the reference has no native code for these; it composes
 * torch.gather/max/softmax -- SURVEY.md section 2b), with the reference's tensor layouts:
 *
 *   ffb6d_random_sample_f32           FFB6D.random_sample           ffb6d/models/ffb6d.py:159-177
 *   ffb6d_nearest_interpolation_f32   FFB6D.nearest_interpolation   ffb6d/models/ffb6d.py:179-194
 *                                     (also the `choose` gather,    ffb6d/models/ffb6d.py:309-312)
 *   ffb6d_gather_neighbour_f32        Building_block.gather_neighbour        RandLANet.py:225-234
 *   ffb6d_relative_pos_encoding_f32   Building_block.relative_pos_encoding   RandLANet.py:216-223
 *   ffb6d_att_pool_f32                Att_pooling.forward softmax/mul/sum    RandLANet.py:245-248
 *   *_bwd_f32                         the autograd of the above (the reference gets it from torch)
 *
 * Conventions: all tensors are contiguous row-major device buffers, features float32;
 * index tensors are int64 (what the model receives, train_lm.py:236-237) or int32 (what
 * the dataset produces, linemod_dataset.py:319-339) selected by `idx_bits` (64 or 32);
 * indices must lie in [0, M) -- out-of-range indices are a caller error and are
 * reported by ffb6d_check_index_range(), the kernels themselves do not bounds-check.
 * Calls are stream-ordered on `stream` and never synchronise.
 * Return: FFB6D_OK or a negative FFB6D_ERR_* (message via ffb6d_last_error()).
 */
#ifndef FFB6D_OPS_H_
#define FFB6D_OPS_H_

#include "ffb6d_knn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* out[b,c,n] = max_k feat[b,c,idx[b,n,k]]           feat [B,C,M], idx [B,Np,K], out [B,C,Np]
 * arg (nullable) [B,C,Np] int32: the winning source column (first maximum in k order,
 * as torch.max returns), used by the backward. */
int ffb6d_random_sample_f32(const float* feat, const void* idx, int idx_bits,
                            float* out, int32_t* arg,
                            int64_t B, int64_t C, int64_t M, int64_t Np, int K,
                            ffb6d_stream_t stream);

/* grad_feat[b,c,arg[b,c,n]] += grad_out[b,c,n]; grad_feat [B,C,M] is zeroed first. */
int ffb6d_random_sample_bwd_f32(const float* grad_out, const int32_t* arg, float* grad_feat,
                                int64_t B, int64_t C, int64_t M, int64_t Np,
                                ffb6d_stream_t stream);

/* out[b,c,u] = feat[b,c,idx[b,u]]                   feat [B,C,M], idx [B,U], out [B,C,U] */
int ffb6d_nearest_interpolation_f32(const float* feat, const void* idx, int idx_bits,
                                    float* out,
                                    int64_t B, int64_t C, int64_t M, int64_t U,
                                    ffb6d_stream_t stream);

/* grad_feat[b,c,idx[b,u]] += grad_out[b,c,u]; grad_feat [B,C,M] is zeroed first. */
int ffb6d_nearest_interpolation_bwd_f32(const float* grad_out, const void* idx, int idx_bits,
                                        float* grad_feat,
                                        int64_t B, int64_t C, int64_t M, int64_t U,
                                        ffb6d_stream_t stream);

/* out[b,n,k,:] = pc[b,idx[b,n,k],:]                 pc [B,M,C], idx [B,N,K], out [B,N,K,C] */
int ffb6d_gather_neighbour_f32(const float* pc, const void* idx, int idx_bits, float* out,
                               int64_t B, int64_t M, int64_t C, int64_t N, int K,
                               ffb6d_stream_t stream);

/* grad_pc[b,idx[b,n,k],:] += grad_out[b,n,k,:]; grad_pc [B,M,C] is zeroed first. */
int ffb6d_gather_neighbour_bwd_f32(const float* grad_out, const void* idx, int idx_bits,
                                   float* grad_pc,
                                   int64_t B, int64_t M, int64_t C, int64_t N, int K,
                                   ffb6d_stream_t stream);

/* out[b,n,k,0:10] = [ |p-q|, p-q, p, q ],  p = xyz[b,n], q = xyz[b,idx[b,n,k]]
 * xyz [B,N,3], idx [B,N,K], out [B,N,K,10];  |.| = sqrt((dx*dx+dy*dy)+dz*dz), f32, no FMA */
int ffb6d_relative_pos_encoding_f32(const float* xyz, const void* idx, int idx_bits, float* out,
                                    int64_t B, int64_t N, int K, ffb6d_stream_t stream);

/* s = softmax_k(act[b,c,n,:]);  out[b,c,n] = sum_k feat[b,c,n,k] * s[k]
 * feat, act [B,C,N,K] (K in {1,2,4,8,16,32}), out [B,C,N] */
int ffb6d_att_pool_f32(const float* feat, const float* act, float* out,
                       int64_t B, int64_t C, int64_t N, int K, ffb6d_stream_t stream);

/* relative_pos_encoding written channel-major: out [B,10,N,K] (what RandLANet.py:198 permutes to). */
int ffb6d_relative_pos_encoding_cm_f32(const float* xyz, const void* idx, int idx_bits, float* out,
                                       int64_t B, int64_t N, int K, ffb6d_stream_t stream);

/* grad_feat = g*s ; grad_act = g*s*(feat - out)  with g = grad_out[b,c,n] broadcast over k */
int ffb6d_att_pool_bwd_f32(const float* grad_out, const float* feat, const float* act,
                           float* grad_feat, float* grad_act,
                           int64_t B, int64_t C, int64_t N, int K, ffb6d_stream_t stream);

/* Valid-pixel sampling + point assembly of Dataset.get_item (linemod_dataset.py:262-289) without a host round trip:
 * per frame a uniformly random N-subset of the pixels with depth > min_depth in uniformly random order (fewer than N valid
 * pixels: their random permutation repeated cyclically), from one batched radix sort of hashed keys.  depth [B,H,W];
 * xyz [B,3,H,W] (ffb6d_depth_to_cloud_f32), rgb [B,3,H,W] uint8 (rgb_is_u8) or float32, nrm [B,3,H,W].  Outputs:
 * choose [B,N] int64 (pixel index y*W+x), cld [B,N,3], cld_rgb_nrm [B,9,N] (ffb6d.py:222-224 input layout),
 * n_valid [B] int32 (valid pixels per frame; the reference drops frames with fewer than 400).  seed: any 64-bit value.
 * xyz, rgb, nrm, cld and cld_rgb_nrm may all be NULL: only choose and n_valid are produced then. */
size_t ffb6d_sample_points_workspace_bytes(int64_t B, int64_t H, int64_t W);
int ffb6d_sample_points_f32(const float* depth, float min_depth, const float* xyz, const void* rgb, int rgb_is_u8,
                            const float* nrm, uint64_t seed, int64_t* choose, float* cld, float* cld_rgb_nrm,
                            int32_t* n_valid, int64_t B, int64_t H, int64_t W, int64_t N, void* workspace,
                            size_t workspace_bytes, ffb6d_stream_t stream);

/* Surface normals of a depth image: the published algorithm of `normalSpeed.depth_normal(dpt_mm, fx, fy, k_size,
 * distance_threshold, difference_threshold, point_into_surface)` (third-party, not vendored in the reference; call sites
 * linemod_dataset.py:252-254, ycb_dataset.py:208-210: (dpt_mm, K[0][0], K[1][1], 5, 2000, 20, False)) = OpenCV LINE-MOD
 * bilateral least-squares normals, see csrc/inputs.hip.  depth_mm [B,H,W]: uint16 (depth_is_u16) or float32 truncated to
 * uint16 like `.astype(np.uint16)`; out [B,3,H,W] float32 unit normals ((0,0,0) where undefined).  Parity unpinned. */
int ffb6d_depth_normal(const void* depth_mm, int depth_is_u16, double fx, double fy, int kernel_size,
                       int distance_threshold, int difference_threshold, int point_into_surface, float* out, int64_t B,
                       int64_t H, int64_t W, ffb6d_stream_t stream);

/* YCB depth hole filling: Basic_Utils.fill_missing(dpt, cam_scale, scale_2_80m) (utils/basic_utils.py:467-487) ->
 * fill_in_multiscale(extrapolate=False, blur_type='bilateral', max_depth) (utils/ip_basic/ip_basic/depth_map_utils_ycb.py:
 * 290-445), restated operator by operator (csrc/holefill.hip).  depth [B,H,W] in the caller's unit, out the filled map in
 * the same unit; the reference's call (ycb_dataset.py:204) is (dpt_um, cam_scale, 1) with max_depth = 3.0.  Parity
 * unpinned (OpenCV absent here).  workspace: ffb6d_fill_missing_workspace_bytes(B, H, W) bytes of device scratch. */
size_t ffb6d_fill_missing_workspace_bytes(int64_t B, int64_t H, int64_t W);
int ffb6d_fill_missing_f32(const float* depth, double cam_scale, double scale_2_80m, float max_depth, float* out, int64_t B,
                           int64_t H, int64_t W, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream);

/* ==== point-major / pixel-major ("channels last") operators: one row of C contiguous floats per point or pixel ====
 * (csrc/mlp_pm.hip, csrc/ops_pm.hip).  Rows run over the points of ALL frames unless stated; row strides (ld*) are in floats. */

/* Shared MLP: out[r,m] = act( sum_k w[m,k] * X[r,k] + bias[m] + y[yrow(r), m] ), X = [x1 | x2] along k.
 * w [cout, k1+k2] is the conv weight in the layout nn.Conv stores it (BN folded); x1 [.., ld1] / x2 [rows, ld2], the first
 * k1 / k2 floats of a row are used (multiples of 8).  Optional operand gather: with x1_idx [rows] the x1 row of output row r
 * is (r / rows_per_frame) * x1_rows_per_frame + x1_idx[r] (the `choose` pick of ffb6d.py:309-312 fused into the head GEMM).
 * Optional epilogue rows y [.., ldy]: with y_idx [rows] row (r / rows_per_frame) * y_rows_per_frame + y_idx[r]
 * (conv(cat(a, interp(b))) == W_a a + gather(W_b b), ffb6d.py:247-253,273-279), with y_idx NULL row r itself.
 * idx_bits 32/64 applies to both index arrays.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 log_softmax over the cout <= 64
 * channels (pspnet.py:108-112).  out [rows, ldo].  tile_hint 0 = choose the kernel form from the problem
 * (ffb6d_mlp_pm_choice). */
int ffb6d_mlp_pm_f32(const float* w, const float* bias, const float* x1, int64_t k1, int64_t ld1, const void* x1_idx,
                     int64_t x1_rows_per_frame, const float* x2, int64_t k2, int64_t ld2, const float* y, int64_t ldy,
                     const void* y_idx, int64_t y_rows_per_frame, int idx_bits, int64_t rows_per_frame, float* out,
                     int64_t ldo, int64_t rows, int64_t cout, int act, int tile_hint, ffb6d_stream_t stream);

/* Tile shape ffb6d_mlp_pm_f32 picks for tile_hint 0 (pure host logic): 1 = 128 ch x 128 pt, 2 = 64 x 256, 3 = 32 x 256,
 * 4 = 64 x 64, 5 = 64 x 32 with K split over the four waves -- i.e. which kernel instantiation a profile will list. */
int ffb6d_mlp_pm_tile(int64_t rows, int64_t cout, int64_t K, int act);
/* Kernel form for tile_hint 0: 6 = the stream form (mlp_pm_stream_kernel: persistent workgroups, W resident in LDS, whole-row
 * loads and stores through wave-private LDS images) for the HBM-bound short-row layers, 7 = the LDS-tiled form
 * (mlp_pm_lds_kernel: 128 x 128 tiles, 128-byte row segments staged through LDS) for the big long-row layers, else
 * ffb6d_mlp_pm_tile's tile.  All forms give identical results. */
int ffb6d_mlp_pm_choice(int64_t rows, int64_t cout, int64_t k1, int64_t k2, int act, int bf16, int x1_gathered);
/* 8 + 256 * T (round 5, fp32): the tile-SEQUENCE form of the LDS-tiled kernel (mlp_pm_seq_kernel) -- a workgroup multiplies T
 * consecutive channel tiles of one point tile and the epilogue of each finished tile (second accumulator set) rides between the
 * MFMAs of the next one; ffb6d_mlp_pm_choice returns it for the long-row fp32 layers with at least two channel tiles per sequence,
 * tile_hint = 8 + 256 * plan selects it explicitly (plan 0: ffb6d_mlp_pm_seq_plan).  plan: bits 0-3 = T of the first region of point
 * tiles, bits 4-7 = T of the second region, bits 8-15 / 16-22 = tiles (in units of 16) of the second region / of the last region, whose
 * workgroups take one tile each -- a guided schedule: the workgroups handed out last are short.  Identical results. */
int ffb6d_mlp_pm_seq_plan(int64_t rows, int64_t cout);
/* Round 6: plan = 0xF0 | rounds (| sequences per XCD << 8): every XCD's tile list (its point tiles, channel tiles fastest) is cut into
 * 64 * rounds contiguous sequences whose lengths differ by at most one, the longer ones first -- a sequence may run from one point tile
 * into the next.  What ffb6d_mlp_pm_seq_plan returns for layers of <= 8 channel tiles; ffb6d_mlp_pm_set_seq_lin(0) = the round-5 plans. */
void ffb6d_mlp_pm_set_seq_lin(int on);
/* 9 (round 6, bf16): the 256 x 256 tile with LDS-DMA operand loads (csrc/mlp_pm_big.hip: `buffer_load_dwordx4 ... lds` into
 * chunk-permuted 128-byte image rows, eight waves, one workgroup per CU) -- ffb6d_mlp_pm_choice returns it for the bf16 launches with
 * K >= 256, cout >= 192 and at least 512 such tiles; ffb6d_mlp_pm_set_big_form(0) keeps the automatic choice on form 7 (A/B).
 * Identical results. */
void ffb6d_mlp_pm_set_big_form(int on);

/* Att_pooling.forward up to the pooled tensor (RandLANet.py:243-248) with the neighbour gather fused in:
 * S[(n,k),:] = [ f[nei[n,k],:] | g[(n,k),:] ], out[n,m] = sum_k S[(n,k),m] * softmax_k((S w_fc^T)[(n,k),m]).
 * f [B*N, ldf] point rows (c1 channels), nei [B,N,16] indices inside the frame, g [B*N*16, ldg] pair rows (c2 channels),
 * w_fc [c1+c2, c1+c2] (conv layout), out [B*N, ldo].  Neither the gathered tensor nor the scores are written. */
int ffb6d_att_pool_pm_f32(const float* w_fc, const float* f, int64_t c1, int64_t ldf, const void* nei, int idx_bits,
                          const float* g, int64_t c2, int64_t ldg, float* out, int64_t ldo, int64_t B, int64_t N, int K,
                          ffb6d_stream_t stream);

/* bf16 twins of the two GEMM-shaped operators (BASELINE.json configuration 5, mixed precision): activations, weights
 * and the epilogue rows y are bfloat16, bias stays float32, accumulation and all epilogue arithmetic are fp32
 * (v_mfma_f32_32x32x16_bf16); k1, k2 / c1, c2 must be multiples of 16. */
int ffb6d_mlp_pm_bf16(const void* w, const float* bias, const void* x1, int64_t k1, int64_t ld1, const void* x1_idx,
                      int64_t x1_rows_per_frame, const void* x2, int64_t k2, int64_t ld2, const void* y, int64_t ldy,
                      const void* y_idx, int64_t y_rows_per_frame, int idx_bits, int64_t rows_per_frame, void* out,
                      int64_t ldo, int64_t rows, int64_t cout, int act, int tile_hint, ffb6d_stream_t stream);
int ffb6d_att_pool_pm_bf16(const void* w_fc, const void* f, int64_t c1, int64_t ldf, const void* nei, int idx_bits,
                           const void* g, int64_t c2, int64_t ldg, void* out, int64_t ldo, int64_t B, int64_t N, int K,
                           ffb6d_stream_t stream);

/* Row operators below take `dtype`: 0 = float32 rows, 1 = bfloat16 rows (channel counts multiples of 4 resp. 8);
 * arithmetic is fp32 either way. */
/* FFB6D.random_sample (ffb6d.py:159-177): out[b,n,:] = max_k feat[b, idx[b,n,k], :]; feat [B,M,C], idx [B,Np,K]. */
int ffb6d_random_sample_pm(int dtype, const void* feat, const void* idx, int idx_bits, void* out, int64_t B, int64_t M,
                           int64_t C, int64_t Np, int K, ffb6d_stream_t stream);
/* FFB6D.nearest_interpolation / the `choose` pick (ffb6d.py:179-194,309-312): out[b,u,:] = feat[b, idx[b,u], :]. */
int ffb6d_gather_rows_pm(int dtype, const void* feat, const void* idx, int idx_bits, void* out, int64_t B, int64_t M,
                         int64_t C, int64_t U, ffb6d_stream_t stream);
/* relative_pos_encoding (RandLANet.py:216-223) as rows of 16 channels: [dis, p-q, p, q, 0 x 6]; xyz fp32, out [B,N,K,16]. */
int ffb6d_relative_pos_encoding_pm(int dtype, const float* xyz, const void* idx, int idx_bits, void* out, int64_t B,
                                   int64_t N, int K, ffb6d_stream_t stream);
/* out = act( scale[c]*x + shift[c] + (res ? (rscale ? rscale[c]*res + rshift[c] : res) : 0) ) on [rows, C]; scale/shift
 * fp32; act: 0 none, 1 ReLU, 2 leaky/PReLU with `slope`.  In place (out == x) allowed. */
int ffb6d_affine_act_pm(int dtype, const void* x, const float* scale, const float* shift, const void* res,
                        const float* rscale, const float* rshift, void* out, int64_t rows, int64_t C, int act, float slope,
                        ffb6d_stream_t stream);
/* Stem of the colour branch in one pass (extractors.py conv1 -> bn1 -> relu -> MaxPool2d(3,2,1); ffb6d.py:222):
 * out[b,oy,ox,:] = max over the 3x3 / stride-2 / pad-1 window (pixels inside the map) of relu(scale*x + shift);
 * x [B,IH,IW,C] rows, out [B,(IH-1)/2+1,(IW-1)/2+1,C]; NaN propagates like torch.max_pool2d. */
int ffb6d_affine_relu_maxpool_pm(int dtype, const void* x, const float* scale, const float* shift, void* out, int64_t B, int64_t IH,
                                 int64_t IW, int64_t C, ffb6d_stream_t stream);
/* Bilinear resize [B,IH,IW,C] -> [B,OH,OW,C] (ATen upsample_bilinear2d arithmetic; pspnet.py:24-28,37-42). */
int ffb6d_bilinear_resize_pm(int dtype, const void* in, void* out, int64_t B, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                             int64_t C, int align_corners, ffb6d_stream_t stream);
/* Three shared MLPs in a row on [rows, 128] float32 rows as one launch (the layers after the first of a prediction head,
 * ffb6d.py:135-157,316-318): out = act3(W3 act2(W2 act1(W1 x + b1) + b2) + b3), W1, W2 [128,128], W3 [cout3,128] with BatchNorm
 * folded; the hidden activations stay in registers.  Weights k-chunked ([K/4][cout][4] floats: element (q, c, j) = W[c][4q + j]),
 * W3 padded with zero rows to 32 output channels (its k-chunked image is [32][32][4] whatever cout3 is; of b3 only the first cout3
 * entries are read); cout3 (a multiple of 4, <= 32) channels of a row are written; act codes as
 * ffb6d_mlp_pm (0 none, 1 ReLU, 2 LeakyReLU(0.2)).  Per output the k order of ffb6d_mlp_pm's tile kernels. */
int ffb6d_mlp_chain3_pm_f32(const float* x, int64_t ldx, const float* w1k, const float* b1, int act1, const float* w2k, const float* b2,
                            int act2, const float* w3k, const float* b3, int act3, float* out, int64_t ldo, int64_t rows, int64_t cout3,
                            ffb6d_stream_t stream);
/* bfloat16 twin (round 5): rows, weights and output bfloat16, biases float32, accumulation and epilogue arithmetic float32, the hidden
 * activations rounded to bfloat16 where the separate launches would store them.  Weights k-chunked in chunks of 8 ([K/8][cout][8]); W1 in
 * the natural k order, W2 and W3 with their chunks PERMUTED to the order the accumulators of the previous layer supply k in: chunk 2m + h
 * (m = 0..7, h = 0..1) holds the input channels base + (0..3), base + 8 + (0..3) with base = 32 (m >> 1) + 16 (m & 1) + 4 h
 * (ffb6d_amd.ops_pm.k_chunked(w, perm=True)). */
int ffb6d_mlp_chain3_pm_bf16(const void* x, int64_t ldx, const void* w1k, const float* b1, int act1, const void* w2k, const float* b2,
                             int act2, const void* w3k, const float* b3, int act3, void* out, int64_t ldo, int64_t rows, int64_t cout3,
                             ffb6d_stream_t stream);
/* PSPUpsample's convolution where its output is read (pspnet.py:34-45; the last colour stage feeds the heads only through the `choose`
 * pick, ffb6d.py:302-312): out [B*P, 9, C] = the 3x3 patches of the align_corners bilinear up-sampling of in [B,IH,IW,C] to OH x OW
 * around the picked pixels idx [B*P] (flat Y*OW + X within the frame; int32 / int64), tap-major, zeros outside the map -- the operand
 * rows of a K = 9*C ffb6d_mlp_pm GEMM with the weight [Cout, (ky*3+kx)*C + ci].  Arithmetic as ffb6d_bilinear_resize_pm. */
int ffb6d_upsampled_patch_rows_pm(int dtype, const void* in, const void* idx, int idx_bits, void* out, int64_t B, int64_t IH,
                                  int64_t IW, int64_t OH, int64_t OW, int64_t C, int64_t P, ffb6d_stream_t stream);
/* Relative position encoding fused with the first shared MLP of the local feature aggregation (Building_block.forward,
 * RandLANet.py:196-199: mlp1(relative_pos_encoding(xyz, neigh_idx)), encoding [dis, p-q, p, q] of RandLANet.py:216-223):
 *   out[b,n,k,:] = act(w[:, 0:10] . enc(b,n,k) + bias),   xyz [B,N,3] float32, idx [B,N,K], w [cout, ldw] float32 with
 * BatchNorm folded (columns 10.. ignored), out [B,N,K,cout] rows of dtype (0 = float32, 1 = bfloat16); act 0/1/2 =
 * none / ReLU / LeakyReLU(0.2).  The 10-channel encoding is never written. */
int ffb6d_posenc_mlp_pm(int dtype, const float* xyz, const void* idx, int idx_bits, const float* w, int64_t ldw, const float* bias,
                        int act, void* out, int64_t B, int64_t N, int K, int64_t cout, ffb6d_stream_t stream);
/* One half of the local feature aggregation as ONE launch (Building_block.forward, RandLANet.py:196-214; relative_pos_encoding
 * :216-223, gather_neighbour :225-234, Att_pooling.forward :243-250).  With e = [|p-q|, p-q, p, q] of the pair (n, k), g1 =
 * act1(w1 e + b1) (lfa.mlp1) and, for mode 2, g2 = act2(w2 g1 + b2) (lfa.mlp2):
 *   S[(n,k),:] = [ f[nei[n,k],:] | g_mode ],   out[n,:] = actm( wm . sum_k S[(n,k),:] * softmax_k((S wfc^T)[(n,k),:]) + bm )
 * mode 1 = gather + mlp1 + att_pooling_1 (fc, softmax, pool, mlp): f [B*N, ldf] with d/2 channels -> out [B*N, ldo], d/2 channels;
 * mode 2 = gather + mlp1 + mlp2 + att_pooling_2: f = the output of mode 1 -> out with d channels.
 * xyz4 = coordinate table of 16-byte rows {x, y, z, -} float32, point n of frame b at row b * xyz_frame_stride + n (a level that
 * is the prefix of a finer one shares its table: stride = the finer level's N); nei [B,N,16] indices inside the frame,
 * w1 [d/2, ldw1 >= 10] / b1 float32 (BatchNorm folded), w2 [d/2, d/2],
 * wfc [d, d] of dtype (0 = float32, 1 = bfloat16 rows), wm_kc = the [cout, d] weight of the output MLP re-laid as
 * [d / VL, cout, VL] (VL = 4 float32 / 8 bfloat16: 16 bytes of consecutive k per channel, so that the per-channel dot products of
 * the output MLP read it coalesced), b2 / bm float32; act* 0/1/2 = none / ReLU / LeakyReLU(0.2);
 * d in {32, 64, 128, 256}, K = 16.  Neither the encoding, nor the per-pair rows, nor the scores, nor the pooled rows touch HBM
 * (pair rows are staged in LDS, csrc/lfa_pm.hip).  p_hint = size + 8 * w: size 0 = automatic, 1 = 1024/d points per group, 2 = 512/d,
 * 3 = 256/d (d <= 64), 4 = one wave per workgroup with 128/d points (d <= 64); w 0 = automatic, 1 = fc / mlp weights resident in
 * LDS (d <= 64), 2 = streamed from L2. */
int ffb6d_lfa_pm(int dtype, int mode, const float* xyz4, int64_t xyz_frame_stride, const void* nei, int idx_bits, const void* f, int64_t ldf,
                 const float* w1, int64_t ldw1, const float* b1, int act1, const void* w2, const float* b2, int act2,
                 const void* wfc, const void* wm_kc, const float* bm, int actm, void* out, int64_t ldo, int64_t B, int64_t N,
                 int K, int64_t d, int p_hint, ffb6d_stream_t stream);
/* the p_hint an automatic launch (p_hint 0) resolves to (pure host logic) */
int ffb6d_lfa_pm_choice(int64_t npts, int64_t d, int bf16);
/* Backward bodies of the colour decoder's PSPUpsample for training (csrc/train_ops.hip), on pixel-major rows = torch's
 * channels_last memory format; dtype 0 = float32, 1 = bfloat16 rows, fp32 arithmetic.
 * ffb6d_bilinear_bwd_pm: gradient of the align_corners = True bilinear up-sampling (pspnet.py:37-42) as a gather with ATen's
 * source-index arithmetic: grad_out [B,OH,OW,C] -> grad_in [B,IH,IW,C] (up-sampling by at most 4, maps of at least 2 x 2 pixels).
 * ffb6d_prelu_fwd / _bwd: single-slope PReLU (pspnet.py:43) on n elements; the backward writes grad_x and the slope's gradient
 * (*grad_slope is overwritten: reduced inside the kernel, fp32). */
int ffb6d_bilinear_bwd_pm(int dtype, const void* grad_out, void* grad_in, int64_t B, int64_t IH, int64_t IW, int64_t OH, int64_t OW,
                          int64_t C, ffb6d_stream_t stream);
int ffb6d_prelu_fwd(int dtype, const void* x, const float* slope, void* y, int64_t n, ffb6d_stream_t stream);
int ffb6d_prelu_bwd(int dtype, const void* x, const void* grad_out, const float* slope, void* grad_x, float* grad_slope, int64_t n,
                    ffb6d_stream_t stream);

/* Neighbour operators of the training step on rows (csrc/train_rows.hip): `dtype` rows as above, channel counts and row strides
 * (ld*, in elements) multiples of 4 (float32) / 8 (bfloat16); forward gathers and the max-pool are ffb6d_gather_rows_pm /
 * ffb6d_random_sample_pm.
 * ffb6d_gather_sum_rows: backward of a row gather (gather_neighbour RandLANet.py:225-234, nearest_interpolation ffb6d.py:179-194,
 *   the `choose` pick ffb6d.py:309-312) with the index inverted by the caller: out[r,:] = sum_{j in [start[r], start[r+1])}
 *   g[order[j],:]; g [rows, ldg], order = the gather's flat output rows sorted by their source row, start [R+1] (both int64),
 *   out [R, C] of `dtype`.
 * ffb6d_random_sample_rows_bwd: backward of FFB6D.random_sample (ffb6d.py:159-177): acc[b, idx[b,n,k*], c] += g[b,n,c] with k* the
 *   first neighbour attaining the maximum of channel c (a NaN wins, as in torch.max); feat [B,M,C], idx [B,Np,K], g [B*Np rows, ldg].
 * ffb6d_att_pool_rows: out[p,:] = sum_k feat[p*K+k,:] * softmax_k(scores[p*K+k,:]) (Att_pooling.forward, RandLANet.py:245-248);
 *   feat / scores [P*K rows, ldf / lds], out [P, C].  _bwd: gfeat = g * s, gscores = g * s * (feat - out), both [P*K, C]. */
int ffb6d_gather_sum_rows(int dtype, const void* g, int64_t ldg, const int64_t* order, const int64_t* start, void* out, int64_t R,
                          int64_t C, ffb6d_stream_t stream);
int ffb6d_random_sample_rows_bwd(int dtype, const void* feat, const void* idx, int idx_bits, const void* g, int64_t ldg, float* acc,
                                 int64_t B, int64_t M, int64_t C, int64_t Np, int K, ffb6d_stream_t stream);
int ffb6d_att_pool_rows(int dtype, const void* feat, int64_t ldf, const void* scores, int64_t lds, void* out, int64_t P, int K,
                        int64_t C, ffb6d_stream_t stream);
int ffb6d_att_pool_rows_bwd(int dtype, const void* g, int64_t ldg, const void* feat, int64_t ldf, const void* scores, int64_t lds,
                            void* gfeat, void* gscores, int64_t P, int K, int64_t C, ffb6d_stream_t stream);
/* LogSoftmax over the channels of [R, C] rows (`final` of the colour decoder, pspnet.py:108-112: nn.LogSoftmax() on a 4-d map acts
 * on dim 1) in the rows' own element type, fp32 arithmetic; C / 4 (float32) resp. C / 8 (bfloat16) a power of two <= 64.
 * _bwd: gx = g - softmax(x) * sum_c g, recomputed from the forward's input x. */
int ffb6d_log_softmax_rows(int dtype, const void* x, void* y, int64_t R, int64_t C, ffb6d_stream_t stream);
int ffb6d_log_softmax_rows_bwd(int dtype, const void* g, const void* x, void* gx, int64_t R, int64_t C, ffb6d_stream_t stream);
/* Second half of the folded up-convolution (PSPUpsample, pspnet.py:34-45: bilinear x2 with align_corners -> Conv2d 3x3,
 * padding 1 -> BatchNorm -> PReLU).  z [B,IH,IW,9,C] holds, per low-resolution pixel and filter tap (ky*3+kx), the channel
 * mixing (BatchNorm scale * W[:, :, ky, kx]) x -- one ffb6d_mlp_pm GEMM with 9*C output channels -- and
 *   out[b,Y,X,:] = prelu(shift + sum_taps [tap inside the OHxOW map] * bilinear_align_corners(z[..., tap, :])(Y+ky-1, X+kx-1))
 * with shift [C] float32 = BatchNorm shift + BatchNorm scale * conv bias, one PReLU slope.  Arithmetic of the blend as
 * ffb6d_bilinear_resize_pm (ATen's upsample_bilinear2d); fp32 accumulation in both precisions. */
int ffb6d_upconv_combine_pm(int dtype, const void* z, const float* shift, float slope, void* out, int64_t B, int64_t IH,
                            int64_t IW, int64_t OH, int64_t OW, int64_t C, ffb6d_stream_t stream);
/* A/B of the kernel form on exact x2 maps: 2 = the measured choice per shape (default: the LDS-staged form -- 8 x 16 output pixels of a
 * 64-byte channel chunk per workgroup, window staged with LDS-DMA loads -- for bfloat16 maps of <= 64 channels, the per-thread 2 x 4
 * block elsewhere, bfloat16 on 4-channel half units); 3 = LDS-staged everywhere; 1 = 2 x 4 block everywhere; 0 = as 1 in fp32, one
 * output pixel per thread in bfloat16 (rounds 2-5).  Identical results. */
void ffb6d_upconv_set_form(int form);
/* All adaptive average pools of `sizes` of x [B,H,W,C] -> float32 [B, sum(s*s), C] (bins of sizes[0] first, row-major in a
 * level).  Two passes (row partial sums, then bins) through a workspace of ffb6d_psp_pool_pm_workspace_bytes(...) bytes. */
size_t ffb6d_psp_pool_pm_workspace_bytes(int64_t B, int64_t H, int64_t C, const int* sizes, int nsizes);
int ffb6d_psp_pool_pm(int dtype, const void* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, const int* sizes,
                      int nsizes, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream);
/* out[b,y,x,:] = sum over levels of the bilinear (align_corners = 0) up-sampling of float32 z [B, sum(s*s), M] to (H,W). */
int ffb6d_psp_prior_sum_pm(int dtype, const float* z, void* out, int64_t B, int64_t H, int64_t W, int64_t M, const int* sizes,
                           int nsizes, ffb6d_stream_t stream);

/* Depth image -> xyz image, the dataset's dpt_2_pcld (linemod_dataset.py:188-199,258-259) on the device:
 * depth [B,H,W] f32 (raw units), K [B,3,3] f64 row-major intrinsics, out [B,3,H,W] f32 (x,y,z planes),
 * invalid depth (<= 1e-8 after scaling) and NaN/Inf -> (0,0,0).  float32/float64 mix as numpy there. */
int ffb6d_depth_to_cloud_f32(const float* depth, const double* K, float cam_scale, float* out,
                             int64_t B, int64_t H, int64_t W, ffb6d_stream_t stream);

/* The point sets of the index pyramid (linemod_dataset.py:299-323) in one launch -- what the dataset code produces with slices:
 *   level_out[k] [B, level_n[k], 3] = the first level_n[k] points of every frame of the cloud (level_n[0] = N: the cloud itself as
 *                                      rows; the coarser levels are prefixes of the once-shuffled cloud, :322-323),
 *   table [B, N, 4] = rows {x, y, z, 0} (16-byte coordinate rows of ffb6d_lfa_pm; may be NULL),
 *   grid_out[g] [B, (H / strides[g]) * (W / strides[g]), 3] = the xyz image dpt_xyz [B,3,H,W] at pixels (y * s, x * s), row-major
 *                                      (:299-311); every stride must be a multiple of the smallest one.
 * The cloud is read through element strides: point-major [B,N,3] = (3 N, 1, 3); the first three channels of cld_rgb_nrm [B,9,N]
 * = (9 N, N, 1).  At most 6 levels and 4 grids.  Pure copies (bit-identical to the slices). */
int ffb6d_pyramid_sets_f32(const float* cloud, int64_t cloud_frame_stride, int64_t cloud_coord_stride, int64_t cloud_point_stride,
                           int64_t B, int64_t N, int n_levels, const int64_t* level_n, float* const* level_out, float* table,
                           const float* dpt_xyz, int64_t H, int64_t W, int n_grids, const int* strides, float* const* grid_out,
                           ffb6d_stream_t stream);

/* Debug helper: number of entries of idx[0:count] outside [0, M) written to *bad (device int32). */
int ffb6d_check_index_range(const void* idx, int idx_bits, int64_t count, int64_t M,
                            int32_t* bad, ffb6d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FFB6D_OPS_H_ */
