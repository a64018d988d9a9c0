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

/* Same pooling with the feature set given as two channel blocks, cat(feat1 [B,C1,N,K], feat2 [B,C2,N,K])
 * (RandLANet.py:204,212) without materialising the concatenation; act [B,C1+C2,N,K], out [B,C1+C2,N]. */
int ffb6d_att_pool2_f32(const float* feat1, int64_t C1, const float* feat2, int64_t C2, const float* act,
                        float* out, int64_t B, int64_t N, int K, ffb6d_stream_t stream);

/* relative_pos_encoding written channel-major: out [B,10,N,K] (what RandLANet.py:198 permutes to). */
int ffb6d_relative_pos_encoding_cm_f32(const float* xyz, const void* idx, int idx_bits, float* out,
                                       int64_t B, int64_t N, int K, ffb6d_stream_t stream);

/* grad_feat = g*s ; grad_act = g*s*(feat - out)  with g = grad_out[b,c,n] broadcast over k */
int ffb6d_att_pool_bwd_f32(const float* grad_out, const float* feat, const float* act,
                           float* grad_feat, float* grad_act,
                           int64_t B, int64_t C, int64_t N, int K, ffb6d_stream_t stream);

/* Fused shared MLP (1x1 conv + folded BatchNorm + activation, pytorch_utils.py:75-129 and
 * RandLA/pytorch_utils.py:35-111) on channel-major activations, fp32 MFMA:
 *   out[b,m,p] = act( sum_k wt[k,m] * X[b,k,p] + bias[m] + ygather[b,m,gidx[b,p]] )
 * X = [x1 ; x2] stacked along k (replaces torch.cat(dim=1) in front of a conv, ffb6d.py:252,261,277);
 * ygather/gidx (both or neither) add a column gather of a pre-multiplied matrix: conv(cat(a,
 * interp(b))) == W_a*a + gather(W_b*b) (ffb6d.py:247-253,273-279,283-289).
 * wt [k1+k2, cout] (transposed weights, BN folded), bias [cout] or NULL, x1 [B,k1,P] / x2 [B,k2,P]
 * with the given batch strides (in floats; rows contiguous), ygather [B,cout,py], gidx [B,P]
 * (idx_bits 32/64), out [B,cout,P].  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).
 * workspace: ffb6d_shared_mlp_workspace_bytes(...) bytes of device scratch (may be NULL when 0). */
int ffb6d_shared_mlp_f32(const float* wt, const float* bias, const float* x1, int64_t k1,
                         int64_t x1_batch_stride, const float* x2, int64_t k2, int64_t x2_batch_stride,
                         const float* ygather, const void* gidx, int idx_bits, int64_t py,
                         int64_t yg_batch_stride, float* out, int64_t out_batch_stride, int64_t B,
                         int64_t cout, int64_t P, int act, void* workspace, size_t workspace_bytes,
                         ffb6d_stream_t stream);
/* Scratch for ffb6d_shared_mlp_f32 with K = k1 + k2 (non-zero only for small per-frame P, where the
 * K loop is split across workgroups and reduced by a second kernel). */
size_t ffb6d_shared_mlp_workspace_bytes(int64_t B, int64_t cout, int64_t K, int64_t P);

/* Attentive pooling with the score GEMM fused in (Att_pooling.forward, RandLANet.py:243-248, up to
 * the pooled tensor): scores = W_fc * S over the feature set S = cat(x1 [B,k1,N,16], x2 [B,k2,N,16]),
 * out[b,m,n] = sum_k S[b,m,n,k] * softmax_k(scores[b,m,n,:]).  wt = W_fc transposed [k1+k2, k1+k2];
 * the [B,d,N,16] score tensor is never written.  K must be 16. */
int ffb6d_att_score_pool_f32(const float* wt, const float* x1, int64_t k1, const float* x2, int64_t k2,
                             float* out, int64_t B, int64_t N, int K, ffb6d_stream_t stream);

/* Bilinear resize of `planes` = B*C independent [IH,IW] float32 images to [OH,OW], the two
 * flavours the colour branch uses: align_corners = 0 (F.upsample(size=...), pspnet.py:24-28) and
 * align_corners = 1 (nn.Upsample(scale_factor=2, align_corners=True), pspnet.py:37-42).
 * Same arithmetic as ATen's upsample_bilinear2d. */
int ffb6d_bilinear_resize_f32(const float* in, float* out, int64_t planes, int64_t IH, int64_t IW,
                              int64_t OH, int64_t OW, int align_corners, ffb6d_stream_t stream);

/* Per-channel affine (+ affine residual) + activation on [B,C,HW] maps, in place allowed:
 *   out = act(scale[c]*x + shift[c] + (res ? rscale[c]*res + rshift[c] : 0));  rscale/rshift NULL = 1/0.
 * The eval-mode BatchNorm + ReLU/PReLU + residual-add glue of the colour branch (extractors.py:49-63,
 * pspnet.py:34-45) in one pass.  act: 0 none, 1 ReLU, 2 leaky/PReLU with `slope`.  HW % 4 == 0. */
int ffb6d_affine_act_f32(const float* x, const float* scale, const float* shift, const float* res,
                         const float* rscale, const float* rshift, float* out, int64_t B, int64_t C,
                         int64_t HW, int act, float slope, ffb6d_stream_t stream);

/* log_softmax over the channel axis of [B,C,HW] (pspnet.py:108-112 `final`: nn.LogSoftmax() on a 4-d
 * tensor acts on dim 1), C in {16,32,64}; in place allowed. */
int ffb6d_channel_log_softmax_f32(const float* x, float* out, int64_t B, int64_t C, int64_t HW,
                                  ffb6d_stream_t stream);

/* Pyramid pooling helpers (pspnet.py:7-31).  psp_pool: all adaptive average pools of `sizes` (<= 4
 * sizes, e.g. 1,2,3,6) of `planes` = B*C [H,W] maps in one pass; out [planes, sum(s*s)] (bins of size
 * sizes[0] first, row-major).  psp_prior_sum: out[plane,y,x] = sum_i bilinear(z_i)(y,x) for maps
 * z [planes, sum(s*s)] in the same packing, align_corners = False, W % 4 == 0. */
int ffb6d_psp_pool_f32(const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                       const int* sizes, int nsizes, ffb6d_stream_t stream);
int ffb6d_psp_prior_sum_f32(const float* z, float* out, int64_t planes, int64_t H, int64_t W,
                            const int* sizes, int nsizes, ffb6d_stream_t stream);

/* Depth image -> xyz image, the dataset's dpt_2_pcld (linemod_dataset.py:188-199,258-259) on the device:
 * depth [B,H,W] f32 (raw units), K [B,3,3] f64 row-major intrinsics, out [B,3,H,W] f32 (x,y,z planes),
 * invalid depth (<= 1e-8 after scaling) and NaN/Inf -> (0,0,0).  float32/float64 mix as numpy there. */
int ffb6d_depth_to_cloud_f32(const float* depth, const double* K, float cam_scale, float* out,
                             int64_t B, int64_t H, int64_t W, ffb6d_stream_t stream);

/* Debug helper: number of entries of idx[0:count] outside [0, M) written to *bad (device int32). */
int ffb6d_check_index_range(const void* idx, int idx_bits, int64_t count, int64_t M,
                            int32_t* bad, ffb6d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FFB6D_OPS_H_ */
