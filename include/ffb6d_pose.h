/* include/ffb6d_pose.h -- C ABI of the gfx950 pose solver: the step that follows FFB6D.forward
 * (SURVEY.md section 8f rank 2).  Replaces, batched over every (frame, object) pair at once:
 *
 *   MeanShiftTorch.fit          ffb6d/utils/meanshift_pytorch.py:27-58
 *   best_fit_transform          ffb6d/utils/pvn3d_eval_utils_kpls.py:28-61
 *   the vote construction, mask selection and centre-cluster filtering of
 *   cal_frame_poses / cal_frame_poses_lm   ffb6d/utils/pvn3d_eval_utils_kpls.py:65-158,220-285
 *
 * All pointers are DEVICE pointers unless stated; every call is asynchronous on `stream`
 * except ffb6d_mean_shift_f32 with check_every > 0 (it polls a convergence flag).
 * Return value: 0 or an FFB6D_ERR_* code (text through ffb6d_last_error()).
 *
 * A "vote set" is a list of 3-D points stored as float4 {x, y, z, bit-cast int32 index of the
 * cloud point that cast the vote}; set g occupies sets[g*set_stride .. +counts[g/sets_per_count]).
 */
#ifndef FFB6D_POSE_H
#define FFB6D_POSE_H

#include <stddef.h>
#include <stdint.h>

#include "ffb6d_knn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Votes of the points selected by a class mask, order preserving (pvn3d_eval_utils_kpls.py:73-75:
 * pred = pcld - offset; :112-136 / :257-275: pred[:, mask == cls_id (& centre labels), :]).
 *   pcld    f32 [B,N,3]        offsets f32 [B,S,N,3]      (S = 1 for centre offsets, n_kps for keypoints)
 *   mask    int32|int64 [B,N]  (mask_bits = 32|64)          keep  u8 [B,N] or NULL (extra AND)
 *   pair p  = (frame_of[p], class_of[p]), p < n_pairs;  writes sets [n_pairs*S, set_stride, 4]
 *   (set index p*S + s) and counts [n_pairs]; set_stride >= N is always sufficient. */
int ffb6d_vote_sets_f32(const float* pcld, const float* offsets, const void* mask, int mask_bits,
                        const unsigned char* keep, const int* frame_of, const int* class_of,
                        int n_pairs, int B, int S, int N, int64_t set_stride,
                        float* sets, int* counts, ffb6d_stream_t stream);

/* MeanShiftTorch(bandwidth, max_iter).fit for G sets at once (meanshift_pytorch.py:27-58).
 * Per set: move every point to the Gaussian-weighted mean of the set until the largest move is
 * < bandwidth*1e-3 or max_iter+1 rounds were made; centre = converged point with the most
 * neighbours within `bandwidth` (lowest index on ties), labels[j] = |point_j - centre| < bandwidth.
 *   centers f32 [G,3]; labels u8 [G,set_stride] or NULL; n_inside i32 [G] or NULL (size of the
 *   winning ball); iters i32 [G] or NULL (rounds made).  Sets with count 0 give centre (0,0,0).
 *   max_count: an upper bound on every count if the caller knows one (sizes the launch grids),
 *   0 = unknown (set_stride is used).
 *   check_every > 0: every that many rounds the host reads back the convergence flags and stops
 *   launching when all sets are done (synchronises the stream); 0: all max_iter+1 rounds are
 *   enqueued (finished sets cost an empty launch) and the call never synchronises. */
size_t ffb6d_mean_shift_workspace_bytes(int G, int64_t set_stride);
/* A/B of the one-workgroup fit: 1 (default) = sets of up to 2048 points on the light form (512 threads, 78 KB of LDS: fits beside
 * other workgroups on a CU), larger ones up to 4096 on the round-5 form; 0 = the round-5 form for all of them.  Identical results. */
void ffb6d_pose_set_fit_form(int form);
/* The first two rounds of the sets of 512 .. 4096 points made chip-wide (128 points per workgroup, results in the workspace), the
 * one-workgroup fits starting from them: 1 (default) = when G <= CUs / 2 (the fits alone would leave half of the chip idle),
 * 2 = always, 0 = never (every round inside the fit).  Identical results (same pair arithmetic). */
void ffb6d_pose_set_fit_spread(int on);
/* Sets of more than 4096 points: 1 (default) = their rounds run chip-wide on (position, multiplicity) lists with exact-duplicate merging
 * until at most 4096 distinct positions are left (typically 8-10 rounds), then the one-workgroup fit continues; 0 = the round-by-round
 * path (max_iter + 1 rounds of count^2 pairs).  Results agree to rounding (equal terms are collected, like in the one-workgroup fit).
 * The call synchronises the stream while such sets are in flight (it reads the list lengths back), whatever check_every says. */
void ffb6d_pose_set_big_form(int form);
int ffb6d_mean_shift_f32(const float* sets, const int* counts, int sets_per_count, int G,
                         int64_t set_stride, int64_t max_count, float bandwidth, int max_iter,
                         int check_every, float* centers, unsigned char* labels, int* n_inside,
                         int* iters, void* workspace, size_t workspace_bytes, ffb6d_stream_t stream);

/* keep[frame_of[p], idx] = labels[p, j] for the j-th vote of set p (one set per pair): turns the
 * centre-cluster labels into a per-point filter for the keypoint votes (:131-134 / :270-273).
 * keep u8 [B,N] must be zero-initialised by the caller. */
int ffb6d_set_labels_to_points(const float* sets, const unsigned char* labels, const int* counts,
                               const int* frame_of, int n_pairs, int64_t set_stride, int N,
                               unsigned char* keep, ffb6d_stream_t stream);

/* Centre-clustering mask filter of cal_frame_poses (:85-108): every foreground point goes to the
 * class whose voted centre is nearest to the point's own centre vote, if that distance is below
 * max_dist[p] (= 0.8 * object radius).  Pairs of frame b are pair_begin[b] .. pair_begin[b+1]-1.
 *   centers f32 [n_pairs,3]; mask/mask_out int32|int64 [B,N]. */
int ffb6d_refine_mask_by_center(const float* pcld, const float* ctr_offsets, const void* mask,
                                int mask_bits, const float* centers, const int* class_of,
                                const int* pair_begin, const float* max_dist, int B, int N,
                                void* mask_out, ffb6d_stream_t stream);

/* best_fit_transform for P problems (:28-61): T[p] = [R|t] (row-major double [P,3,4]) minimising
 * sum |R*model[p,i] + t - found[p,i]|^2, reflection-corrected.  model/found f32 [P,n,3].
 * Computed in double from the float inputs. */
int ffb6d_best_fit_transform_f32(const float* model, const float* found, int P, int n, double* T,
                                 ffb6d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
