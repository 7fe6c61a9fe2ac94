/* coponerf_hip.h — C ABI of libcoponerf_hip.so (gfx950 / MI355X).
 *
 * The upstream reference (cvlab-kaist/CoPoNeRF) has no FFI: its hot path is a
 * chain of ATen calls inside models/CoPoNeRF.py:208-576.  Each entry point
 * below replaces one group of those calls; the Python class
 * coponerf_amd.CoPoNeRF.CoPoNeRF (same constructor / get_z / forward contract
 * as models/CoPoNeRF.py:19-576) binds them through ctypes (coponerf_amd/_hip.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; nothing is
 *     allocated or freed by the library; outputs are caller-allocated.
 *   - return value: 0 = ok, otherwise a negative argument error (CPN_E_*) or a
 *     positive hipError_t; cpn_last_error() gives a message.  Never throws.
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream).
 *   - fp16 buffers are IEEE binary16 ("half"), declared as uint16_t here.
 *   - N = B*V camera slots, ordered n = b*V + v  (V == 2).
 *   - "sample arrays" are (N,R,S,·) ; "row arrays" feed the GEMMs and are ordered
 *       row = (((b*R + r)*V + v)*S + s)*J + j        J = 2 encoder inputs per sample
 *     so that all V*S*J rows of one query ray are contiguous.
 */
#ifndef COPONERF_HIP_H
#define COPONERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPN_ABI_VERSION 10

#define CPN_E_ARG   (-1)   /* bad argument (null pointer, size, alignment) */
#define CPN_E_SHAPE (-2)   /* shape not supported by the compiled tiles    */

/* camera block: CPN_CAM_STRIDE floats per (b,v), built on the host from 4x4 matrices */
#define CPN_CAM_STRIDE 96
#define CPN_CAM_TQ    0    /* 16: query camera -> frame of context view v   (CoPoNeRF.py:241-244) */
#define CPN_CAM_M     16   /* 16: inv(c2w_v) @ c2w_v, ~identity             (CoPoNeRF.py:239)     */
#define CPN_CAM_AOWN  32   /* 16: view-v frame -> view-v frame              (CoPoNeRF.py:326-332) */
#define CPN_CAM_AOTH  48   /* 16: view-v frame -> frame of the other view                          */
#define CPN_CAM_KQ    64   /* fx fy cx cy of the query camera, pixel units                         */
#define CPN_CAM_KC    68   /* fx fy cx cy of context view v                                        */
#define CPN_CAM_KO    72   /* fx fy cx cy of the other context view                                */
#define CPN_CAM_KN    76   /* 9: K_v[:3,:3] with rows 0,1 divided by H      (CoPoNeRF.py:259-261) */

/* padded K of the first encoder layer: 835 inputs -> 864 consumed (27 x 32); row stride 896 halves = 14 x 128 B:
 * 128-byte-aligned rows are worth 10 % on the GEMM that reads them (808 vs 728 TFLOP/s, measured in round 1 with a leading-dimension sweep of tools/gemm_bench.py) */

int         cpn_abi_version(void);
const char* cpn_last_error(void);

/* ---- CU-partitioned streams -------------------------------------------------------------------------------------
 * The reference's evaluation loop runs get_z and the chunked forward() calls strictly one after the other
 * (test.py:164-212, wrapper.py:176-211).  On an MI355X the render pass (HBM-bound, persistent grids on every CU) and
 * get_z (hundreds of small launches) only overlap when each has CUs of its own: a stream created here is restricted
 * to CUs [first_cu, first_cu + num_cus) of the CU-mask order, which KFD spreads round-robin over the 8 XCDs and their
 * 4 shader engines — both numbers must be multiples of 32, i.e. an equal share of every shader engine of every XCD.  The persistent launchers of this library
 * (cpn_encode_key, cpn_gemm_f16*) size their grids by cpn_stream_cu_count(stream).  `stream_out` receives a
 * hipStream_t; destroy it with cpn_stream_destroy once its work has completed.  (coponerf_amd/streams.py, pipeline.py) */
int cpn_device_cu_count(void);
int cpn_stream_cu_count(void* stream);
int cpn_stream_create_cu_range(int first_cu, int num_cus, void** stream_out);
int cpn_stream_destroy(void* stream);

/* ---- K1: query rays -> Pluecker coords + clipped epipolar segment ------------------------------
 * replaces geometry.plucker_embedding (utils_training/geometry.py:236-245, :426-433, :409-419, :353-371),
 * epipolar.project_rays (models/epipolar.py:175-253) and the start/end scrub (models/CoPoNeRF.py:279-291).
 *   cam     (N, CPN_CAM_STRIDE)        uv: pixel (x = column, y = row) of ray r of pair b at uv[b*uv_batch_stride + 2r]
 *                                      (2R for a dense (B,R,2) array; a ray chunk of a larger array passes its stride)
 *   coords9 (N,R,9)  = dir3 | moment3 | origin3
 *   seg     (N,R,4)  = start.xy | end.xy in [-1,1], NaN/Inf -> 0
 *   overlaps(N,R)    uint8 0/1                                                                   */
int cpn_project_rays(const float* cam, const float* uv, long long uv_batch_stride, int B, int V, int R,
                     float* coords9, float* seg, uint8_t* overlaps, void* stream);

/* ---- K1b: per-sample geometry -------------------------------------------------------------------
 * replaces sample generation (CoPoNeRF.py:294-309), geometry.get_3d_point_epipolar / get_intersection
 * (geometry.py:98-162, float64 inside), utils.encode_relative_point (utils.py:99-108), geometry.project +
 * utils.normalize_for_grid_sample (geometry.py:374-393, utils.py:242-245), geometry.get_ray_directions_cam
 * (geometry.py:313-324) and the depth encoding (CoPoNeRF.py:428-445).
 *   interval (S) = linspace(0,1,S) as the host computes it
 *   pixel_val (N,R,S,2)   pt (N,R,S,3)   sec_grid (N,R,S,2): coords of this sample's 3-D point in the OTHER image
 *   pe6 (N,R,S,6) = tanh(nan_to_num(pt_own)/5) | tanh(nan_to_num(pt_other)/5)
 *   loc8 (N,R,S,8) = context-pixel ray dir 3 | tanh(depth*{1,.1,.01,.001}) 4 | 0
 *   lv_u, optional (NULL: not written): (B * ceil(R/4) * V * ceil(S/4) units, 64 lanes, 4) fp32 - the 16 local_coords inputs of
 *       every sample (loc8's 7 values, 6 of coords9, a 1.0 for the bias, zeros; CoPoNeRF.py:411-445) in the UNIT order of
 *       cpn_local_units: lane c + 16 fg of unit ((b * ceil(R/4) + r/4) * V + v) * ceil(S/4) + s/4, c = (s & 3) * 4 + (r & 3),
 *       holds K entries 4 fg .. 4 fg + 3 = {dir3, 1} | {0, 0, c9[0], c9[1]} | {c9[2], tanh(depth * {1, .1, .01})} |
 *       {tanh(depth / 1000), c9[6..8]}.  Slots of rays >= R / samples >= S are written as zeros.                               */
int cpn_sample_geometry(const float* cam, const float* coords9, const float* seg, const float* interval,
                        int B, int V, int R, int S, int H, int W,
                        float* pixel_val, float* pt, float* sec_grid, float* pe6, float* loc8, float* lv_u, void* stream);

/* ---- layout: (N,C,h,w) fp32 -> (N,h,w,C) fp16 feature map (once per get_z) --------------------- */
int cpn_nchw_to_nhwc_f16(const float* src, uint16_t* dst, int N, int C, int h, int w, void* stream);

/* ---- pack a (N_out, K_in) fp32 weight into fp16 [N_out][ld] with zero padding (once per weight version) */
int cpn_pack_weight_f16(const float* src, int n_out, int k_in, uint16_t* dst, int ld, void* stream);

/* ---- small fp32 first layers feeding the attention MLPs --------------------------------------------
 * out[row, 0:128] = fp16( relu( W[:, 0:16] . L(row) + bias + add[ray, 0:128] ) ),  rows = (b,r,v,s), ray = (b,r)
 * with L = [loc8 dir 3 | 0 0 0 | query dir 3 | loc8 depth 4 | query origin 3]      (CoPoNeRF.py:445)
 * query_embed (CoPoNeRF.py:446): w = query_embed.weight (128,16), add = NULL
 * query_repeat_embed (CoPoNeRF.py:472-473): w = weight[:, 128:144] (ld 144), add = weight[:, :128] . z_embed + 0 */
int cpn_local_hidden(const float* loc8, const float* coords9, const float* w, int ldw, const float* bias,
                     const float* add, int B, int V, int R, int S, int ray0, int nrays, uint16_t* out, void* stream);

/* ---- K2+K3a: first encoder layer straight from the feature maps ("project, then interpolate") -------------------
 * hid = ReLU(query_encode_latent([primary/secondary gather (832) | tanh(pt/5) (3)])) without the gathered rows ever
 * reaching HBM; replaces F.grid_sample x 8, torch.cat and the 835 -> 832 1x1 convolution (CoPoNeRF.py:312, 370,
 * 384-397 with the layer of :71).  A 1x1 convolution commutes with bilinear interpolation, and the texel centres of the
 * three 256-channel levels (H/16, H/8, H/4; align_corners=False) all lie on the nodes of one grid of spacing 2/W, on
 * whose cells every level's interpolant is bilinear — so the projected sum of the three levels is tabulated once per
 * stereo pair on that grid and interpolated from 4 nodes (exact in real arithmetic; csrc/encode.hip):
 *   per image  T_border (H/2+1, W/2+1, CPN_TAB_LD) fp16   nodes 0..M         own image, 'border' padding
 *              T_zeros  (H/2+9, W/2+9, CPN_TAB_LD) fp16   nodes -4..M+4      other image, 'zeros' padding
 *   laid out image by image, border table first: cpn_encode_table_nodes(H, W) nodes (rows) per image.
 *   Built as  cpn_node_features (the three levels sampled at every node -> (nodes, 768) fp16)
 *             -> cpn_gemm_f16(A = node features, W = wtab, N = CPN_TAB_LD, K = 768, bias 0, fp16 out).
 * A row is then 4 weighted table taps + a K = 80 MFMA product over [level-3 gather (64) | pt (3) | bias hi, lo | 0].
 * Table rows hold the 832 channels in natural order (1664 B, 13 cache lines; the kernel walks 13 slices of 64 channels).
 *   cpn_pack_encode_weights: W (832, ldw >= 835) fp32 query_encode_latent.weight ->
 *       wfrag (13*3*4*64*8 halves) MFMA A-operand fragments of W[:, 768:835] (K padded to 96)
 *       wtab  (832, 768) fp16 table projection weights W[:, 0:768]
 *   the layer itself: cpn_encode_key below (until round 5 also as a kernel without the key layer, cpn_encode_hidden)  */
#define CPN_TAB_LD    832
#define CPN_NODE_PAD  4
long long cpn_encode_table_nodes(int H, int W);
int cpn_pack_encode_weights(const float* W, int ldw, uint16_t* wfrag, uint16_t* wtab, void* stream);
int cpn_node_features(const uint16_t* map0, const uint16_t* map1, const uint16_t* map2, int H, int W, int nimg,
                      uint16_t* out, void* stream);
/* cpn_encode_key (csrc/encode_fused.hip; round 5 form of round 4's kernel): the first layer with the folded key_map layer
 * behind it (models/CoPoNeRF.py:404-407 after :387-397; folding: DESIGN.md 4.3) - the 64-channel slices of hid are the K panel
 * of the 1664 -> 128 contraction while they are still in registers, so the key path never reads hid back from HBM.
 *   kwring (2 images, 13 slices, 8 tiles, 2 k steps, 64 lanes = row + 16 * 8-column group, 8) fp16: the folded key matrix
 *       (Wk_a W2 | Wk_b W2) (128, 1664) in the order the kernel streams it through its two-slot LDS ring - piece (t, k) of slice
 *       step (j, n) holds, in lane (a, g), W'[16 t + a][832 j + 64 n + 32 k + 8 g .. +8]: every 1 KiB DMA piece is contiguous;
 *       kbias (128) fp32 = Wk_a b2 + Wk_b b2 + bk
 *   hid (rays*V*S*2, 832) fp16 in the row order of this header (written: the two hidden sums read it); bias (832) fp32; map3 the
 *       full-resolution NHWC fp16 map (N, H, W, 64); kh (rays*V*S, 128) fp16 = ReLU(W' . [hid_own ; hid_other] + kbias), rows in
 *       the order of this header; kh is bit-identical to cpn_gemm_f16(hid, W', relu).
 *   kh_units = 1: kh leaves in UNIT order instead of row-major - the 16 rows of a unit (4 adjacent rays x 4 consecutive samples of
 *       one view; unit u of a launch = ((ray group - first group) * V + view) * ceil(S/4) + sample block, cpn_encode_units() of them)
 *       as [unit][32-column block p][lane = c + 16 fg][8] = kh[row(unit, c)][32 p + 8 fg .. +8], c = (sample & 3) * 4 + (ray & 3):
 *       every store of a wave is 1 KiB of contiguous memory and every block is a ready MFMA B fragment for cpn_local_units.
 *       The buffer holds cpn_encode_units() * 16 rows (dead rows of partial units included).                                */
int cpn_encode_key(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                   const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                   const uint16_t* kwring, const float* kbias, int B, int V, int R, int S, int ray0, int nrays,
                   uint16_t* hid, uint16_t* kh, int kh_units, void* stream);
long long cpn_encode_units(int B, int R, int S, int ray0, int nrays);
/* ---- the per-sample query / key tails of the two attention rounds in UNIT order (round 5, csrc/local_units.hip) ------------
 * mode 0 (CoPoNeRF.py:408, 446, 450): ce = query_embed_2(ReLU(query_embed(local_coords))) -> ce_u (unit order, cpn_encode_units()
 *   * 16 rows); logits[row] = < fp16(key_map_2(kh_u[row])), fp16(ce[row]) >, kh_u = cpn_encode_key's unit-order output.
 *   w1 (128, ldw1 >= 16) fp32 / b1: first layer; w2 (128, ldw2) fp16 / b2: second layer;
 *   wk2 (128, ldwk2) fp16 / bk2: key_map_2.
 * mode 2 (:472-475): logits[row] = < fp16(query_repeat_embed_2(ReLU(w1 . local_coords + b1 + add[ray]))), ce[row] >, add (nrays,
 *   128) fp32, with coords_embed RECOMPUTED from the local coordinates (the same instructions in the same order as mode 0 formed
 *   it with).  logits (nrays*V*S) fp32 in row order; rows of a partial unit outside the ray range are skipped.  w1b (128, ldw1b >= 16) fp32 / b1b = query_embed, wk2 / bk2 =
 *   query_embed_2; ce_u is not touched.  Mode 0 with ce_u = NULL then stores no coords_embed at all (2 x 256 bytes per sample
 *   less HBM traffic; both kernels were bound by it).  w1b / b1b are ignored by mode 0.  (Mode 1 - round 2 reading a stored
 *   coords_embed - left the library in round 6.)
 *   lv_u, optional (NULL: the kernel reads loc8 / coords9 itself): the unit-order copy of the rows' 16 first-layer inputs that
 *   cpn_sample_geometry writes for the WHOLE (B, R, S) problem - every mode then reads ONE coalesced 1 KiB line per unit instead
 *   of five scattered 4 - 16 byte accesses per lane (the same values: the same logits); the launch's units start at its first
 *   ray group inside it.                                                                                                    */
int cpn_local_units(int mode, const float* loc8, const float* coords9, const float* w1, int ldw1, const float* b1,
                    const float* add, const uint16_t* w2, int ldw2, const float* b2, const uint16_t* wk2, int ldwk2,
                    const float* bk2, const float* w1b, int ldw1b, const float* b1b, const uint16_t* kh_u, int B, int V,
                    int R, int S, int ray0, int nrays, uint16_t* ce_u, const float* lv_u, float* logits, void* stream);

/* ---- K3: fused GEMM  C = act(A . W^T + bias), fp16 in, fp32 accumulate (MFMA 16x16x32 f16) -----------
 * replaces the per-sample 1x1 convolutions (CoPoNeRF.py:387-397, 404, 408, 446, 473).
 *   A (M, lda) fp16, W (N, ldw) fp16 (both K-contiguous), bias (N) fp32, K multiple of 32, N multiple of 16*tile
 *   out_f32 = 0: C fp16 (M, ldc) ; 1: C fp32 (M, ldc) ; 2: C fp32 += the product + bias, then the ReLU (round 6)        */
int cpn_gemm_f16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                 void* C, int ldc, int M, int N, int K, int relu, int out_f32, void* stream);
/* Few-row form (round 4): the per-RAY value projection of a small call (M = 3 641 rays when the callers render an image as 18
 * forward() calls, test.py:176-190) - fp32 out, results bit-identical to cpn_gemm_f16 (same accumulation order), 13 us where
 * the 256-row tiles take 37.  Wp = the weights in MFMA fragment order, written once per parameter version by
 * cpn_pack_gemm_frags: (N/16, K/32, 64 lanes, 8) fp16, lane l of fragment (t, ks) = W[16 t + (l & 15)][32 ks + 8 (l >> 4) .. +8]. */
int cpn_pack_gemm_frags(const uint16_t* W, int ldw, int N, int K, uint16_t* out, void* stream);
int cpn_gemm_f16_fewrows(const uint16_t* A, int lda, const uint16_t* Wp, const float* bias, float* C, int ldc, int M, int N,
                         int K, int relu, void* stream);

/* ---- K4': the same joint softmax, reducing the 1664 hidden activations [h_own ; h_other] of every sample ----
 * (value projection folded through query_encode_latent_2 and applied once per ray afterwards)
 *   hid (rays*V*S, 1664) fp16 = the (rows*2, 832) output of the first encoder layer; hbar (rays, 1664) fp16   */
/*   logits (rays*V*S) fp32 or NULL: the row dot products <qa[row], qb[row]> when the producing kernel already
 *   formed them (cpn_local_units); then qa, qb may be NULL                                                     */
int cpn_attend_hidden(const uint16_t* qa, const uint16_t* qb, const float* logits, const uint16_t* hid, int B, int V,
                      int R, int S, int ray0, int nrays, uint16_t* hbar, float* at_wt, void* stream);

/* ---- K5: exact-fp32 per-ray linear layer (MFMA 16x16x4 f32)  Y = act_out( act_in(X) . W^T + bias + res ) --
 * replaces nn.Conv1d encode_latent (CoPoNeRF.py:468) and lightfield.ResnetFC (models/lightfield.py:131-167).
 *   X (M, ldx), W (N, ldw), bias (N) or NULL, res (M, ldr) or NULL, Y (M, ldy); N <= 128; K multiple of 4      */
int cpn_linear_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* res, int ldr,
                   float* Y, int ldy, int M, int N, int K, int relu_in, int relu_out, void* stream);

/* ---- the light-field decoder with the output masking, ONE launch (round 3) ---------------------------------
 * replaces lightfield.ResnetFC.forward (models/lightfield.py:131-167; ResnetBlockFC :52-61) on
 * [coords of view 0 | coords of view 1 | z_local ; z_local] (models/CoPoNeRF.py:547-560) and the white background of
 * rays no context view sees (:562-566).  Exact fp32 (v_mfma_f32_16x16x4_f32), same operation order as the
 * layer-by-layer cpn_linear_f32 chain + masking pass it superseded on the inference path.
 *   coords9 (N,R,9)   z_local (B*R,416)   overlaps (N,R) uint8
 *   wpack: every weight matrix W[N][K] in MFMA FRAGMENT order (round 4) - [N/16][K/16][64 lanes][4]: lane l of fragment
 *          (t, kb) holds W[16 t + (l & 15)][16 kb + 4 (l >> 4) .. +4], so that a wave's load of one fragment is 1 KiB of
 *          contiguous memory (row-major weights read in that layout cost 64 L1 tag look-ups per instruction) - biases plain:
 *          CPN_LIGHTFIELD_PACK_FLOATS floats = lin_in W[128][32] (18 columns used, rest zero) b[128] |
 *          3 x { lin_z W[128][416] (the two 416-column halves of the reference's 832 summed) b[128] |
 *                fc_0 W[128][128] b[128] | fc_1 W[128][128] b[128] } | lin_out W[16][128] (rows 3.. zero) b[16]
 *   rgb (B,1,R,3)   valid (B,R,1) float 0/1   rgb_raw (B*R,3) or NULL: the decoder output before masking        */
#define CPN_LIGHTFIELD_PACK_FLOATS (128 * 32 + 128 + 3 * (128 * 416 + 128 + 2 * (128 * 128 + 128)) + 16 * 128 + 16)
int cpn_lightfield_decode(const float* coords9, const float* z_local, const float* wpack, const uint8_t* overlaps,
                          int B, int V, int R, float* rgb, float* valid, float* rgb_raw, void* stream);

/* ---- auxiliary per-ray outputs, ONE launch (round 3) -----------------------------------------------------
 * replaces models/CoPoNeRF.py:493-541: at_wt.argmax, the attention-weighted expected 3-D point
 * (clamp(pt,-100,100), summed over samples and views), its depth in the query camera
 * (utils_training/geometry.py:395-406), batch_project_to_other_img into both context views
 * (utils_training/utils.py:140-170), generate_mask_from_confidence_score (:260-276) and flow2kps (:52-69).
 *   at_wt (N,R,S)   pt (N,R,S,3)   uv: pixel (x,y) of ray r of pair b at uv[b*uv_batch_stride + 2 r]
 *   rayc (B, CPN_RAYC_STRIDE): row 2 of inv(query cam2world) (4) | inv(K_query[:3,:3]) (9) | K_ctx0 (9) | K_ctx1 (9) |
 *        Tq[:,0] (16) | Tq[:,1] (16)      (O(B) host pose algebra, like the camera block)
 *   mask2 (B,256,256) uint8/bool: cycle-consistency mask of view 2; flow_up (B,2,256,256): flow[1] at 256x256
 *   at_wt_max (N,R) int64   depth_ray (B,R) clamped to [0,10]   t_to_c1, t_to_c2, c2_to_c1 (B,R,2)
 *   mask_c2, match_mask (B,R) uint8 0/1                                                                         */
#define CPN_RAYC_STRIDE 64
int cpn_ray_outputs(const float* at_wt, const float* pt, const float* uv, long long uv_batch_stride, const float* rayc,
                    const uint8_t* mask2, const float* flow_up, int B, int V, int R, int S, long long* at_wt_max,
                    float* depth_ray, float* t_to_c1, float* t_to_c2, uint8_t* mask_c2, uint8_t* match_mask,
                    float* c2_to_c1, void* stream);

/* ==== the reference-arithmetic mode in the restructured formulation (csrc/encode_f32.hip, round 6) ==================
 * RenderEngine.precision = "f32": fp32 node tables, fp32 blends and products; the hidden activations are carried as fp16
 * (hi, lo) pairs - hs (rows, 3328) fp16 = [hi_own 832 | hi_other 832 | lo_own 832 | lo_other 832], x = hi + lo to 22 bits -
 * so that the folded key layer runs on cpn_gemm_f16 against (hi, lo) weights with exact products
 * (C = [hi | lo] . [W_hi | W_hi]^T, then C += hi . W_lo^T: cpn_gemm_f16 with out_f32 = 2 accumulates onto C).              */

/* the three coarse levels (NHWC fp32 maps (nimg, H/16.., 256)) sampled at every table node (grid_sample semantics of the
 * node's table: 'border' / 'zeros', models/CoPoNeRF.py:312, 370) -> feat (nimg * cpn_encode_table_nodes, 768) fp32         */
int cpn_node_features_f32(const float* map0, const float* map1, const float* map2, int H, int W, int nimg, float* feat,
                          void* stream);

/* first encoder layer on fp32 tables: per sample row (ray, view, sample, image j)
 *   hid = ReLU(sum_t a_t tab[node_t] + W[:, 768:835] . [bilinear(map3) 64 | tanh(pt/5) 3] + b)     (CoPoNeRF.py:384-397)
 * tab (nimg * nodes, 832) fp32 = node features . W[:, :768]^T, map3 (nimg, H, W, 64) fp32 NHWC, w80t (68, 832) fp32 = the
 * layer's columns 768..834 transposed (k-major) with the bias as row 67; output hs (nrays*V*S, 3328) fp16 (hi, lo) pairs  */
int cpn_encode_hidden_f32(const float* tab, const float* map3, int H, int W, const float* pixel_val, const float* sec_grid,
                          const float* pe6, const float* w80t, int B, int V, int R, int S, int ray0, int nrays, uint16_t* hs,
                          void* stream);

/* joint softmax over the V*S samples of a ray of <qa, qb> / 11.31 (fp32 (rows,128) operands) and the weighted sum of the
 * hidden activations hs (hi, lo pairs) -> hbar (nrays, 1664) fp32, at_wt (N,R,S) fp32 or NULL   (CoPoNeRF.py:450-461, 475-485) */
int cpn_attend_hidden_f32(const float* qa, const float* qb, const uint16_t* hs, int B, int V, int R, int S, int ray0, int nrays,
                          float* hbar, float* at_wt, void* stream);

/* ==== training: backward of the two non-GEMM stages (plain GEMM gradients use hipBLASLt via torch.matmul) ==== */

/* gradient of cpn_attend_hidden: dhbar (rays,1664) fp32, dw_ext (N,R,S) fp32 or NULL (external gradient on the
 * softmax weights), at_wt (N,R,S) the forward weights -> dqa, dqb (rays*V*S,128) fp16, dhid (rays*V*S,1664) fp16 or
 * NULL (the caller then forms the hidden-activation gradient of all consumers with cpn_hid_grad_combine).
 * dqb_acc (rays*V*S,128) fp16 or NULL: added to dqb (qb = coords_embed feeds both attention rounds, models/CoPoNeRF.py:450,475:
 * the second call sums the two gradients instead of autograd adding two 0.5 GB tensors)                            */
int cpn_attend_hidden_bwd(const uint16_t* qa, const uint16_t* qb, const uint16_t* hid, const float* at_wt,
                          const float* dhbar, const float* dw_ext, int B, int V, int R, int S, int ray0, int nrays,
                          uint16_t* dqa, uint16_t* dqb, uint16_t* dhid, const uint16_t* dqb_acc, void* stream);

/* gradient w.r.t. the pre-activation of the first encoder layer from ALL consumers of hid, in one pass:
 *   out[row,c] = hid[row,c] > 0 ? dkey[row,c] + w1[n,r,s]*dh1[ray, j*832+c] + w2[n,r,s]*dh2[ray, j*832+c] : 0
 * dkey (rows, 832) fp16 or NULL = gradient through the key path; (w_i (N,R,S) fp32, dh_i (rays,1664) fp32) or NULL =
 * the attention-weighted hidden sums (w_i: forward softmax weights, dh_i: gradient of the summed vector).            */
int cpn_hid_grad_combine(const uint16_t* dkey, const uint16_t* hid, const float* w1, const float* dh1, const float* w2,
                         const float* dh2, int B, int V, int R, int S, int ray0, int nrays, uint16_t* out, void* stream);

/* the same result WITHOUT the stored key-path gradient (round 6): the data-gradient GEMM of the folded key map
 * (autograd's grad_input of Conv2d 1664->128, models/CoPoNeRF.py:404 under wrapper.py:138) with cpn_hid_grad_combine as its
 * epilogue,   out (rows,1664) = mask(hid) . (dkh (rows,K) . Wt (1664,K)^T + w1 (x) dh1 + w2 (x) dh2),   rows = nrays*V*S.
 * dkh fp16 (row stride lda), Wt fp16 = the folded key-map weight transposed (K-contiguous rows, stride ldw), K % 32 == 0,
 * S % 16 == 0.  Same arithmetic as cpn_gemm_f16 followed by cpn_hid_grad_combine (bit-identical), 14 GB less HBM traffic
 * per training step at 4 x 4096 rays.                                                                                     */
int cpn_gemm_f16_combine(const uint16_t* dkh, int lda, const uint16_t* Wt, int ldw, const uint16_t* hid, const float* w1,
                         const float* dh1, const float* w2, const float* dh2, int B, int V, int R, int S, int ray0,
                         int nrays, int K, uint16_t* out, void* stream);

/* data gradient THROUGH a ReLU in one kernel:  out (M,N) fp16 = mask (M,N) > 0 ? A (M,K) . Wt (N,K)^T : 0  — grad_input of a
 * 1x1 Conv2d whose input `mask` is a ReLU output (key_map -> ReLU -> key_map_2, models/CoPoNeRF.py:404-408: autograd's
 * threshold_backward pass over the (rows,128) gradient is gone).  M %% 16 == 0, K %% 32 == 0, N %% 208 == 0 or N %% 128 == 0. */
int cpn_gemm_f16_masked(const uint16_t* A, int lda, const uint16_t* Wt, int ldw, const uint16_t* mask, int ldm, uint16_t* out,
                        int ldc, int M, int N, int K, void* stream);

/* weight gradients of the 128-wide per-sample layers (torch.mm(dY.t(), X) in the reference's autograd, i.e. the
 * Conv2d(128->128,1x1) / Conv2d(16->128,1x1) layers of models/CoPoNeRF.py:82,85-86,95-96 under wrapper.py:138):
 *   dW (128,128) fp32 += dY^T . X,  db (128) fp32 += column sums of dY (or NULL);  dY (M,128) fp16, X (M,ldx) fp16 of
 *   which the first 128 columns are used.  Both outputs are accumulated: the caller zeroes them.                      */
int cpn_wgrad_skinny_f16(const uint16_t* dY, const uint16_t* X, int ldx, long long M, float* dW, float* db, void* stream);
/* weight gradient of the 832-wide first encoder layer (dW of query_encode_latent, models/CoPoNeRF.py:437-438 under autograd):
 *   dW (N, K) fp32 = dY^T . X / scale[0]  (scale NULL -> 1), written;  dY (M, ldy) fp16 of which N columns are used, X (M, ldx)
 *   fp16 of which K columns are used;  N % 208 == 0, K % 128 == 0 (832 x 896 on the render path), 16-byte aligned rows.
 *   part: cpn_wgrad_tall_scratch(N, K) floats (per-row-slab partial sums, added in a fixed order: deterministic).        */
long long cpn_wgrad_tall_scratch(int N, int K);
int cpn_wgrad_tall_f16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, long long M, int N, int K, const float* scale,
                       float* part, float* dW, void* stream);
/* backward of cpn_local_hidden in one pass: ds, out (rows,128) fp16 (incoming gradient carrying `scale[0]`, a device
 * scalar, and the forward output for the ReLU mask), loc8 / coords9 as in the forward -> dW (128,16), db (128) fp32
 * ACCUMULATED (caller zeroes), dadd (B*R,128) fp32 written (per-ray sum, NULL when the layer had no `add`).          */
int cpn_local_hidden_bwd(const uint16_t* ds, const uint16_t* out, const float* loc8, const float* coords9,
                         const float* scale, int B, int V, int R, int S, float* dW, float* db, float* dadd, void* stream);

/* gradient of the bilinear gather of the full-resolution level w.r.t. its map (F.grid_sample backward, CoPoNeRF.py:312, 370;
 * no coordinate gradient, :380-381): the level's 64 gradient columns start at column col0 of dxin (rows, ldx) fp16; dmap3
 * (N,H,W,64) fp32 NHWC accumulated (caller zeroes); chunk_boxes: scratch of B*V*cpn_gather_bwd_chunks(R,S)*16 int32.     */
long long cpn_gather_bwd_chunks(int R, int S);
int cpn_gather_rows_bwd_level3(const uint16_t* dxin, int ldx, int col0, int H, int W, const float* pixel_val,
                               const float* sec_grid, int B, int V, int R, int S, int ray0, int nrays, float* dmap3,
                               int32_t* chunk_boxes, void* stream);

/* ---- backward of the first encoder layer in its table form (round 3, csrc/encode_bwd.hip) ---------------------------------
 * The layer is hid = ReLU(sum_t a_t T[node_t] + W[:,768:835].[gather_3 | tanh(pt/5)] + b) with T = node_features . W[:,:768]^T
 * (replaces autograd through models/CoPoNeRF.py:312, 370, 384-397 for training).  d (rows, ldx >= 832) fp16 = gradient
 * of the pre-activation (ReLU mask applied, carrying the pass's power-of-two scale).
 *   cpn_scatter_rows_tables: dtab (N * cpn_encode_table_nodes(H,W), 832) fp32, ZERO on entry, += a_t * d[row] at the four
 *       nodes of every row (own image -> border table, other image -> zeros table; the taps of cpn_encode_key).
 *       scratch: cpn_scatter_tables_scratch(H,W,B,V,R,S) int32, 16-byte aligned (bucket counters, work list and one
 *       16-byte descriptor per (row, tile it touches): the rows are counting-sorted by 8x4-node tile first).
 *   cpn_node_features_bwd: adjoint of cpn_node_features: dfeat (nodes, 768) fp32 -> dmap0..2 (N,h,w,256) fp32 NHWC,
 *       OVERWRITTEN (every texel is written once: a gather over the nodes whose footprint holds it, no atomics).
 *   cpn_gather_tail: xt (rows, 128) fp16 = [bilinear gather of the full-resolution map (64) | tanh(pt/5) (3) | 1 | 0 x 60],
 *       the K = 80 operand of the forward kernel with a ones column for the bias gradient.
 *   cpn_scale_to_f16: y = fp16(x * s) with s = 2^floor(log2(target / max|x|)) clamped to [2^-40, 2^40] found on the device and
 *       written to scale_out[0], 1 / s to scale_out[1] (TWO floats; the table gradient sums up to thousands of rows per node:
 *       its own scale before the two fp16 GEMMs; the trunk's convolution backward scales every layer's incoming gradient
 *       this way); amax_scratch: one uint32, ZERO on entry; n % 4 == 0.                                             */
long long cpn_scatter_tables_scratch(int H, int W, int B, int V, int R, int S);
int cpn_scale_to_f16(const float* x, long long n, float target, uint32_t* amax_scratch, uint16_t* y, float* scale_out,
                     void* stream);
int cpn_scatter_rows_tables(const uint16_t* d, int ldx, int H, int W, const float* pixel_val, const float* sec_grid, int B,
                            int V, int R, int S, int ray0, int nrays, float* dtab, int32_t* scratch, void* stream);
int cpn_node_features_bwd(const float* dfeat, int H, int W, int nimg, float* dmap0, float* dmap1, float* dmap2,
                          void* stream);
int cpn_gather_tail(const uint16_t* map3, int H, int W, const float* pixel_val, const float* sec_grid, const float* pe6,
                    int B, int V, int R, int S, int ray0, int nrays, uint16_t* xt, void* stream);

/* ==== get_z path: the 4-D operators of UFC =========================== */

/* ---- K6: Conv4d (+ MaxPool4d when stride > 1) + GroupNorm(1 group) + ReLU in one pass --------------
 * replaces conv4d.Conv4d / MaxPool4d / Encoder4D (models/conv4d.py:7-30, 57-163) and the einops rearrange copies
 * around them.  x (B,Cin,Hq,Wq,Hs,Ws) fp32; wq/ws (Cout,Cin,k,k), bq/bs (Cout): query / support 2-D kernels;
 * y (B,Cout,Hq',Wq',Hs',Ws') with n' = (n + 2p - k)/s + 1; gn_w/gn_b (Cout) GroupNorm affine;
 * stats: cpn_gn_stats_doubles(B, Cout, npos') float64, ZERO on entry.  On return stats[2b], stats[2b+1] = sum / sum of
 * squares of sample b's (Cout, volume) slab; the rest is the arrival counters and per-workgroup partial pairs of the
 * deterministic reduction (the last workgroup of a sample to finish sums the partials in a fixed order: two runs give
 * the same bits).  cpn_gn_relu / cpn_gn_relu_bwd read the first 2B entries.
 * scratch: cpn_conv4d_scratch(...) floats for the pooled volumes of a strided layer (0 for stride 1; NULL = no scratch,
 * the pooling window is then re-evaluated per tap).                                                             */
long long cpn_conv4d_scratch(int B, int Cin, int Hq, int Wq, int Hs, int Ws, int s);
long long cpn_gn_stats_doubles(int B, int Cout, long long npos);
/* residual (same shape as y) or NULL: added to the block's output in the normalisation pass, y = residual + ReLU(GN(conv)) —
 * the `x + Encoder4D(...)` of models/aggregation.py:306, 347-355 without a separate add over the volume.              */
int cpn_conv4d_gn_relu(const float* x, const float* wq, const float* bq, const float* ws, const float* bs,
                       const float* gn_w, const float* gn_b, const float* residual, float eps, int B, int Cin, int Cout,
                       int Hq, int Wq, int Hs, int Ws, int k, int s, int p, float* y, double* stats, float* scratch,
                       void* stream);

/* the two halves of K6 on their own (training keeps the pre-normalisation volume; the data gradient of a stride-1
 * Conv4d is a Conv4d with flipped, transposed kernels): conv (+pool) -> y and the GroupNorm sums; y -> out           */
int cpn_conv4d(const float* x, const float* wq, const float* bq, const float* ws, const float* bs, int B, int Cin,
               int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s, int p, float* y, double* stats, float* scratch,
               void* stream);
/* x (N, P, Q) fp32 -> y (N, Q, P): the (query, support) pair swap of a 4-D volume, (B,C,Hq,Wq,Hs,Ws) ->
 * (B,C,Hs,Ws,Hq,Wq) with N = B*C, P = Hq*Wq, Q = Hs*Ws (x.permute(0,1,4,5,2,3).contiguous(), models/aggregation.py:349) */
int cpn_transpose_pairs(const float* x, int N, int P, int Q, float* y, void* stream);

/* data gradient of the k3 s1 p1 Conv4d (the layer's two (Cout,Cin,3,3) filters read in place, transposed and flipped):
 * dy (B,Cout,Hq,Wq,Hs,Ws) -> dx (B,Cin,Hq,Wq,Hs,Ws); Cin % 4 == 0                                                  */
int cpn_conv4d_dgrad(const float* dy, const float* wq, const float* ws, int B, int Cout, int Cin, int Hq, int Wq, int Hs,
                     int Ws, float* dx, void* stream);
int cpn_gn_relu(const float* y, const double* stats, const float* gn_w, const float* gn_b, const float* residual, float eps,
                int B, int C, long long npos, float* out, void* stream);
/* backward of GroupNorm(1 group) + ReLU (autograd of models/conv4d.py:150-158): y pre-normalisation volume, out the
 * forward output, dout its gradient, stats the forward sums; red (B*2 + C*2) float64 scratch, zero on entry ->
 * dy (B,C,npos), dgn_w (C), dgn_b (C)                                                                              */
int cpn_gn_relu_bwd(const float* y, const float* out, const float* dout, const double* stats, const float* gn_w,
                    float eps, int B, int C, long long npos, double* red, float* dy, float* dgn_w, float* dgn_b,
                    void* stream);

/* weight gradient of ONE separable branch of a 3x3 / stride-1 / pad-1 Conv4d (autograd of models/conv4d.py:108-135):
 * x (B,Cin,G,H,W), dy (B,Cout,G,H,W), convolution over (H,W) -> dw (Cout,Cin,3,3) and db (Cout, may be NULL), both
 * overwritten.  The support branch is the call on (B,C,Hq*Wq,Hs,Ws) as stored; the query branch the call on the
 * volumes with the index pairs swapped.  Cin, Cout <= 32, H*W <= 256, W >= 4.
 * partial: scratch of cpn_conv_wgrad_scratch(Cin, Cout) floats (per-workgroup partial sums).                        */
long long cpn_conv_wgrad_scratch(int Cin, int Cout);
int cpn_conv_wgrad_planes(const float* x, const float* dy, int B, int Cin, int Cout, int G, int H, int W,
                          float* partial, float* dw, float* db, void* stream);

/* weight / bias gradient of the depthwise 3x3 / stride-1 / pad-1 convolution of the UFC feed-forward blocks
 * (DWConv, models/aggregation.py): x, dy (N,C,H,W) -> dw (C,1,3,3), db (C, may be NULL), overwritten.              */
int cpn_dwconv3x3_wgrad(const float* x, const float* dy, int N, int C, int H, int W, float* dw, float* db,
                        void* stream);

/* the same DWConv directly on the token layout x (B, H*W, C) fp32 the feed-forward blocks work in (the reference
 * transposes to (B,C,H,W) around a grouped Conv2d, aggregation.py:18-28): y = bias + w (*) x (flip = 0, forward) or the
 * data gradient w_flipped (*) dy (flip = 1, bias NULL).  w (C,9).  C % 4 == 0.  flip | 2: the exact GELU that follows the
 * DWConv in the feed-forward block (aggregation.py:180, nn.GELU()) is applied to y in the same pass (inference).       */
int cpn_dwconv3x3_tokens(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int flip, float* y,
                         void* stream);
/* its weight / bias gradient: dw (C,9), db (C, may be NULL), overwritten; partial = scratch of
 * cpn_dwconv3x3_tokens_wgrad_scratch(B,H,C) floats (per-row partial sums, reduced in a fixed order)                  */
long long cpn_dwconv3x3_tokens_wgrad_scratch(int B, int H, int C);
int cpn_dwconv3x3_tokens_wgrad(const float* x, const float* dy, int B, int H, int W, int C, float* partial, float* dw,
                               float* db, void* stream);

/* dual softmax of the pose branch's cross attention (models/backbone.py:296-330), a (B,L,M) fp32:
 * f = softmax(a, dim=-1) * softmax(a, dim=-2); rstat (B,L,2) / cstat (B,M,2) receive (max, sum exp) per row / column
 * and feed the backward: da from df with srow (B,L), scol (B,M) scratch.                                            */
int cpn_dual_softmax(const float* a, int B, int L, int M, float* rstat, float* cstat, float* f, void* stream);
int cpn_dual_softmax_bwd(const float* a, const float* rstat, const float* cstat, const float* f, const float* df,
                         int B, int L, int M, float* srow, float* scol, float* da, void* stream);

/* ---- K7: cosine correlation of two token sets ------------------------------------------------------
 * replaces aggregation.correlation / correlation_token (models/aggregation.py:70-80):
 * out[b,s,t] = <src[b,s]/(|src[b,s]|+eps), trg[b,t]/(|trg[b,t]|+eps)>; src, trg (B,L,C), C % 16 == 0;
 * src_n, trg_n (B,L,C) scratch for the normalised tokens; out (B,L,L) == (B,1,h,w,h,w).                    */
int cpn_correlation(const float* src, const float* trg, int B, int L, int C, float eps,
                    float* src_n, float* trg_n, float* out, void* stream);
/* VJP of the row normalisation of K7 (training): x, y = x / (|x| + eps), dy (rows, C) fp32, C <= 1024 ->
 * dx = dy / (|x| + eps) - y (y . dy) / |x|, one launch (the stock-op form was ten per call, 26 calls per step).       */
int cpn_l2norm_rows_bwd(const float* x, const float* y, const float* dy, long long rows, int C, float eps,
                        float* dx, void* stream);

/* ---- weight / bias gradient of an fp32 Linear layer (training, round 4) ---------------------------------
 * replaces the dW = dY^T . X and db = dY.sum(0) autograd performs for torch.nn.functional.linear
 * (models/aggregation.py:200-260 projection and feed-forward layers of the cost aggregation; models/CoPoNeRF.py:468 and
 * models/lightfield.py:52-61,131-167 per-ray layers): the tokens R are the contraction index of both row-major operands.
 *   dY (R, O) row stride ldy   X (R, I) row stride ldx   ->   dW (O, I) contiguous, db (O) or NULL
 *   O, I, ldy, ldx multiples of 4, operands 16-byte aligned; exact fp32 products (v_mfma_f32_16x16x4_f32), row slabs summed
 *   in a fixed order (deterministic); scratch: cpn_wgrad_f32_scratch_floats(R, O, I) floats                              */
long long cpn_wgrad_f32_scratch_floats(long long R, int O, int I);
int cpn_wgrad_f32(const float* dY, int ldy, const float* X, int ldx, long long R, int O, int I, float* dW, float* db,
                  float* scratch, void* stream);

/* ---- the Adam update of every parameter tensor, ONE launch (training, round 4) ------------------------------
 * replaces torch.optim.Adam.step() in the reference's loop (train.py:102-105, wrapper.py:149-151): default betas / eps, no
 * weight decay, no amsgrad; fp32, the operation order of the library's fused kernel.
 *   segs: device array of CPN_ADAM_SEG_BYTES-byte records, one per tensor, little-endian:
 *         { float* param; const float* grad (NULL: tensor skipped); int64 offset of its moments in exp_avg / exp_avg_sq
 *           (multiple of 4); int32 numel; float step_size = lr / (1 - beta1^k); float 1 / sqrt(1 - beta2^k); 12 bytes pad }
 *   blocks: device array of nblocks x { int32 tensor, int32 first element }: cpn_adam_chunk() elements per block
 *   gscale: device scalar multiplied into every gradient first (the clip coefficient), or NULL
 *   gate: device scalar, 0 = skip the whole update (the finite-gradient guard of wrapper.py:44-58,147-151 without a host
 *         read), or NULL.  counts_in / counts_out (one int32 per tensor, two DIFFERENT arrays swapped by the caller every
 *         step; or both NULL): the tensors' update counts kept on the device — the kernel then forms step_size and
 *         1 / sqrt(1 - beta2^k) from counts_in[t] + 1 and `lr` itself and ignores the two floats of the record             */
#define CPN_ADAM_SEG_BYTES 48
int cpn_adam_chunk(void);
int cpn_adam_step(const void* segs, const int* blocks, int nblocks, float* exp_avg, float* exp_avg_sq,
                  const float* gscale, const float* gate, const int* counts_in, int* counts_out, double lr,
                  double beta1, double beta2, double eps, void* stream);

/* ---- K8: soft-argmax with temperature over the 4-D correlation, both directions --------------------
 * replaces aggregation.soft_argmax + softmax_with_temperature (models/aggregation.py:119-144, 555-560).
 * c (B, h*h source, h*h target); t_to_s[b,:,s] = E_{t ~ softmax_t(c[b,s,:]/beta)}[(x_t, y_t)],
 * s_to_t[b,:,t] = E_{s ~ softmax_s(c[b,:,t]/beta)}[(x_s, y_s)], coordinates linspace(-1,1,h); outputs (B,2,h,h). */
int cpn_soft_argmax_pair(const float* c, int B, int h, float beta, float* t_to_s, float* s_to_t, void* stream);
/* backward of K8: t_to_s / s_to_t are the forward outputs, g_* their gradients (B,2,h*h) -> dc (B,h*h,h*h), overwritten */
int cpn_soft_argmax_pair_bwd(const float* c, int B, int h, float beta, const float* t_to_s, const float* s_to_t,
                             const float* g_t_to_s, const float* g_s_to_t, float* dc, void* stream);

/* ---- K9: linear attention of UFCLayer.forward_attention (models/aggregation.py:84-117) ------------------------------
 * phi = ELU + 1;  out = phi(Q) . (sum_s phi(K)_s (x) V_s / L) * L / (phi(Q) . sum_s phi(K)_s + eps)
 *   q, k (B, L, H, 32) fp32;  v and out: channel_major = 0 -> (B, L, H, Dv) (feature branch),
 *   channel_major = 1 -> (B, H, Dv, L) (cost-volume branch: (B, H*Ht*Wt, fs, fs) maps as stored, no permute copies)
 *   scratch: cpn_linear_attention_scratch(B, H, Dv, nsplit) floats; nsplit = number of token ranges reduced in
 *   parallel (partials are summed in a fixed order: deterministic)                                             */
long long cpn_linear_attention_scratch(int B, int H, int Dv, int nsplit);
int cpn_linear_attention(const float* q, const float* k, const float* v, int B, int L, int H, int Dv,
                         int channel_major, float eps, int nsplit, float* scratch, float* out, void* stream);
/* VJP of K9 (autograd through models/aggregation.py:84-117 in the reference): dout has out's layout, dv has v's;
 * dq, dk (B, L, H, 32).  scratch: cpn_linear_attention_bwd_scratch(B, L, H, Dv, nsplit) floats.  All three are overwritten. */
long long cpn_linear_attention_bwd_scratch(int B, int L, int H, int Dv, int nsplit);
int cpn_linear_attention_bwd(const float* q, const float* k, const float* v, const float* dout, int B, int L, int H, int Dv,
                             int channel_major, float eps, int nsplit, float* scratch, float* dq, float* dk, float* dv,
                             void* stream);

/* ---- q / k of UFCLayer.forward_attention (models/aggregation.py:276-281) from the low-resolution projection (round 3):
 *   q | k = lin + upsample(low, fs x fs, bilinear, align_corners=True) + pos_embed
 * lin (B, fs*fs, 2d): the feature half of [q_proj | k_proj] with the biases; low (B, 2d, h, w): the cost-volume half of the
 * two projections applied at the volume's native h x w positions (a Linear layer commutes with the interpolation of
 * aggregation.py:274: 6 - 16x fewer FLOPs); pos (fs*fs, dim); d = nhead*dim.  q, k (B, fs*fs, nhead, dim).          */
int cpn_qk_assemble(const float* lin, const float* low, const float* pos, int B, int fs, int h, int w, int nhead, int dim,
                    float* q, float* k, void* stream);

/* ---- cost-volume side of UFCLayer.forward_attention at the volume's native resolution (round 3) ----------------------
 * replaces models/aggregation.py:283-297 (interpolate value_corr to fs x fs, LinearAttention, interpolate the message
 * back) and the residual of :301: out = residual + D (LinearAttention(q, k, U v_low)), formed as (D diag(Z) phi(q)) .
 * ((U^T phi(k))^T v_low) — U / D the bilinear up / down-sampling (align_corners=True), exact in real arithmetic.
 *   q, k (B, fs*fs, H, 32)   v_low, residual (may be NULL), out (B, H, hs*hs, Dv)
 *   scratch: cpn_cost_volume_attention_scratch(B, fs*fs, H, hs*hs, Dv) floats                                          */
long long cpn_cost_volume_attention_scratch(int B, int L, int H, int P, int Dv);
int cpn_cost_volume_attention(const float* q, const float* k, const float* v_low, const float* residual, int B, int fs,
                              int H, int hs, int Dv, float eps, float* scratch, float* out, void* stream);

/* ---- VJP of the strided Conv4d layers (models/conv4d.py:57-135 with stride > 1: the max-pooled branches; k3 s2 p1 on
 * 32^4 and k5 s4 p2 on 64^4 volumes in UFC.embedding / UFCLayer.feat_to_corr1,2, aggregation.py:198-199, 369-371).
 * Replaces autograd through two max_pool2d + two conv2d + their permute copies.  x (B,Cin,Hq,Wq,Hs,Ws), dy the gradient of
 * the layer's output (B,Cout,Oq,Pq,Os,Ps) BEFORE GroupNorm, wq / ws (Cout,Cin,k,k).  dx (like x) or NULL; gwq, gws
 * (Cout,Cin,k,k) and gb (Cout: the gradient of EITHER bias) or all three NULL.  Maxima are routed to the first maximum of a
 * window in scan order, NaN wins (max_pool2d's rule).  Deterministic.  scratch: cpn_conv4d_strided_bwd_scratch(...) floats.
 * Compiled for Cout = 8, Cin in {1, 2, 8}, k <= 7.                                                                                */
long long cpn_conv4d_strided_bwd_scratch(int B, int Cin, int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s, int p);
int cpn_conv4d_strided_bwd(const float* x, const float* dy, const float* wq, const float* ws, int B, int Cin, int Cout,
                           int Hq, int Wq, int Hs, int Ws, int k, int s, int p, float* scratch, float* dx, float* gwq,
                           float* gws, float* gb, void* stream);

/* ---- K10: cost-volume cross attention of UFCLayer.forward_cross (models/aggregation.py:327-328) -------------------
 * corr (B, H, S, T) fp32; src_v (B, S, H, C), trg_v (B, T, H, C), C == 32
 *   src_attn (B, S, H, C) = softmax over t of corr . trg_v ;  trg_attn (B, T, H, C) = softmax over s of corr, transposed . src_v */
int cpn_cross_attention(const float* corr, const float* src_v, const float* trg_v, int B, int H, int S, int T, int C,
                        float* src_attn, float* trg_attn, void* stream);
/* VJP of K10 (autograd through models/aggregation.py:327-328): g_src / g_trg = gradients of src_attn / trg_attn (the forward
 * outputs, passed back in); dcorr (B,H,S,T), dsrc_v (B,S,H,C), dtrg_v (B,T,H,C) are written.
 * scratch: cpn_cross_attention_bwd_scratch(B, H, S, T) floats.  C == 32, S, T <= 512.                                      */
long long cpn_cross_attention_bwd_scratch(int B, int H, int S, int T);
int cpn_cross_attention_bwd(const float* corr, const float* src_v, const float* trg_v, const float* src_attn,
                            const float* trg_attn, const float* g_src, const float* g_trg, int B, int H, int S, int T, int C,
                            float* scratch, float* dcorr, float* dsrc_v, float* dtrg_v, void* stream);

/* ---- f3: conv_map, the 7x7 3 -> 64 convolution behind the full-resolution feature level (CoPoNeRF.py:69, 182-187) -----
 * rgb (N, H, W, 3) fp32 in [-1, 1] exactly as the input dict holds it; fused (rgb+1)/2, ImageNet normalisation
 * (utils_training/utils.py:247-257), zero-padded 7x7 convolution and bias.  w (64, 3, 7, 7), bias (64).
 * out_nchw (N, 64, H, W) fp32 = z[3] of get_z; out_nhwc_f16 (N, H, W, 64) fp16 or NULL: the layout the render path gathers from */
int cpn_conv_map7x7(const float* rgb, const float* w, const float* bias, int N, int H, int W, float* out_nchw,
                    uint16_t* out_nhwc_f16, void* stream);

/* ---- f3: inference BatchNorm (+ residual) (+ ReLU) of the ResNet-34 trunk (models/backbone.py:10-102; torchvision's
 * BasicBlock: bn(conv(x)), `out += identity`, relu) in one pass over an NCHW fp32 map:
 *   y = act((x - mean[c]) / sqrt(var[c] + eps) * w[c] + b[c] + res),  res (N,C,HW) or NULL, relu 0/1, HW % 4 == 0; y may be x */
int cpn_bn_act(const float* x, const float* res, const float* mean, const float* var, const float* w, const float* b,
               float eps, int N, int C, int HW, int relu, float* y, void* stream);

/* ---- f3: the deep trunk layers as split-K implicit GEMMs on the fp32 MFMA with the BatchNorm / residual / ReLU epilogue
 * (inference, round 4): replaces conv + bn (+ identity) (+ relu) of torchvision's BasicBlock in layer3 / layer4 of the
 * ResNet-34 trunk (models/backbone.py:10-102) at one or a few stereo pairs, where the library's kernels leave most CUs idle.
 *   x (N, Hin, Win, Cin) NHWC fp32, Cin % 16 == 0;  wp = cpn_pack_conv_weight(w (Cout, Cin, k, k)) = [tap][ci][co], Cout % 4 == 0
 *   k in {1, 3} (padding k / 2), stride in {1, 2};  mean / var / bn_w / bn_b (Cout), eps: inference batch norm
 *   res (N, Hout, Wout, Cout) NHWC or NULL, relu 0/1;  out_nhwc and / or out_nchw (N, Cout, Hout, Wout), at least one
 *   scratch: cpn_trunk_conv_scratch_floats(...) floats.  Exact fp32 products, fixed summation order (bit-reproducible).   */
int cpn_pack_conv_weight(const float* w, int Cout, int Cin, int ksize, float* wp, void* stream);
long long cpn_trunk_conv_scratch_floats(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride);
int cpn_trunk_conv_bn_act(const float* x, const float* wp, int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride,
                          const float* mean, const float* var, const float* bn_w, const float* bn_b, float eps,
                          const float* res, int relu, float* out_nhwc, float* out_nchw, float* scratch, void* stream);

/* ---- f1: the launch-bound ends of the pose head, one launch each (inference, round 4) -----------------------------
 * cpn_pose_positional: (x^2, y^2, xy, x, y, 1) of the K^-1-normalised n x n grid (models/backbone.py:209-278) from the raw
 *   intrinsics (B, V, 4, 4) of the input dict (view 0, divided by H like CoPoNeRF.py:199-201); lin = linspace(-1, 1, n);
 *   out (B, n*n, 6), index = col * n + row.
 * cpn_pose_tail: pose_regressor[1:] -> [:, :128] -> rotation / translation regressors -> 6-D rotation -> rel_pose (B, 4, 4)
 *   (models/CoPoNeRF.py:106-126,190-206) from h512 = pose_regressor[0](pose_feat) (B, 512), BEFORE its ReLU.
 *   weights: HOST array of CPN_POSE_TAIL_TENSORS device pointers, weight then bias of each Linear in the order
 *   pose[2] (256 x 512), pose[4] (256 x 256), rotation 128->64->32->6, translation 128->64->32->3.
 *   nsplit > 0: h512 holds the chunk partials (B, 512, nsplit) of cpn_pose_gemv instead and bias0 (512) is pose[0].bias.
 * cpn_pose_gemv: pose_regressor[0] without its bias at B <= 4 rows — partial[b][o][c] = <W[o], x[b]> over chunk c of the
 *   K = (16*16 + 6) * 256 * 2 inputs (W (O, K) fp32, read once for all rows; K % 4 == 0, nsplit <= 64).                    */
#define CPN_POSE_TAIL_TENSORS 16
int cpn_pose_positional(const float* intrinsics, int B, int V, float H, const float* lin, int n, float* out, void* stream);
int cpn_pose_gemv(const float* x, const float* W, int B, int K, int O, int nsplit, float* partial, void* stream);
int cpn_pose_tail(const float* h512, int nsplit, const float* bias0, const float* const* weights, int B, float* rel_pose,
                  void* stream);

/* ---- bilinear resize, align_corners=True, of `planes` independent (h,w) fp32 images -> (H,W) ----------
 * replaces F.interpolate(..., mode='bilinear', align_corners=True) in interpolate4d / forward_attention /
 * interpolate2d_token (models/aggregation.py:49-63, 285, 293, 299).                                           */
int cpn_resize_bilinear_ac(const float* src, float* dst, long long planes, int h, int w, int H, int W, void* stream);
/* its adjoint (the VJP autograd needs): g on the (H,W) grid -> out on the (h,w) grid, gathered in a fixed order (deterministic) */
int cpn_resize_bilinear_ac_adjoint(const float* g, float* out, long long planes, int h, int w, int H, int W, void* stream);

/* ---- UFC.forward's final correlation (models/aggregation.py:549-553): mean of the three levels' correlations after
 * interpolate4d (aggregation.py:49-56: bilinear, align_corners=True, over the target pair of dims, then the source pair)
 * to the finest grid.  c0 (B,1,h0,h0,h0,h0), c1 (B,1,h1,h1,h1,h1), c2 and out (B,1,n,n,n,n) fp32; one pass, in the
 * arithmetic order of the separate resize / add / divide kernels.                                                   */
int cpn_corr_mean3(const float* c0, int h0, const float* c1, int h1, const float* c2, int n, int B, float* out,
                   void* stream);

/* ==== input pipeline: uint8 frames -> the float tensors of the input dict ========================
 * replaces the host-side square crop + `rgb.astype(np.float32) / 127.5 - 1` + query-pixel selection of
 * data/realestate10k_dataio.py:333-441 (utils_training/data_util.py:116-121).
 *   frames_u8 (B, 3, Hs, Ws, 3) uint8: context view 0, context view 1, query frame of every sample
 *   crop window rows [y0, y0+H), columns [x0, x0+W);  ray_pix (B, R) int32 = y*W + x in the cropped query frame
 *   ctx_rgb (B, 2, H, W, 3) fp32, qry_rgb (B, 1, R, 3) fp32: u8 / 127.5 - 1, bit-identical to the reference's numpy  */
int cpn_prepare_input(const uint8_t* frames_u8, int B, int Hs, int Ws, int y0, int x0, int H, int W, int R,
                      const int32_t* ray_pix, float* ctx_rgb, float* qry_rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COPONERF_HIP_H */
