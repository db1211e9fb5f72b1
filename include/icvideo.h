/*
 * icvideo.h — C ABI of libicvideo.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * buffer-conditioned Wan2.1 DiT denoising loop behind InfiniCube's
 * `infinicube.videogen.WanVideoGenerator`.
 *
 * The reference has NO FFI for this path: its boundary is a Python class that forwards to the
 * third-party `diffsynth` package [R infinicube/videogen/inference.py:25-26,216-226].  Each entry
 * point below therefore cites the reference call site whose work it performs ("replaces") and
 * the row of SURVEY.md §8(a-3) (K1..K13) that specifies its math.  The Python binding a
 * maintainer adds is the ctypes stub in INTEGRATION.md (infinicube_amd/native.py is that stub).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - Every tensor pointer is a BORROWED DEVICE pointer (HBM); the caller owns the memory.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it.
 *   - Return 0 on success; non-zero = error, text via icv_last_error() (thread-local).
 *   - bf16 = raw uint16 storage; "f32" = float.  Row-major everywhere; ld* are in ELEMENTS.
 *   - head_dim is fixed at 128 (both Wan2.1 sizes).
 */
#ifndef ICVIDEO_H_
#define ICVIDEO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: icv_unpatchify_cfg_euler gained `round_bf16` (a signature change a host built against 1 cannot detect otherwise);
 *    icv_dit_set_fp8 / icv_dit_set_seqpar and the e4m3 / sequence-parallel bind names were added. */
/* 3: icv_ipc_* (the copy-engine K|V transport), icv_conv3d_ndhwc and the padded-volume VAE helpers were added; no existing
 *    signature changed.
 * 4: icv_attention_fwd_pieces (ONE arrival-gated attention launch per layer over K|V pieces), icv_ipc_arrival / _gather_consumed / _configure / _check /
 *    _drain / _probe_copy, icv_probe_copy_path (arrival flags, bounded device-side waits, teardown that does not depend on live peers, copy-engine-or-blit
 *    probe), icv_flag_write; no existing
 *    signature changed. */
#define ICV_ABI_VERSION 5

/* ---- library / device ------------------------------------------------------------------ */
int icv_abi_version(void);
const char* icv_last_error(void);
/* out[0]=multiprocessor (CU) count, out[1]=max LDS bytes per block, out[2]=gcn arch as int (950),
 * out[3]=warp (wavefront) size.  Replaces nothing in the reference (device probing). */
int icv_device_info(int device, int64_t out[4]);

/* Kernel-variant switches for A/B measurement; defaults = shipped configuration.  "gemm256" = 0 | 1 | 2 (heuristic),
 * "gemm256_mfma" = 16 | 32, "gemm256_sched" = bit 0: two 32-MFMA phases per K-tile, bit 1: batched residual loads (default 3;
 * 7 = + B1 requested a full tile ahead, 11 = + serpentine MFMA order: both measured ties; + 16 / + 32 = non-temporal DMA of the
 * activation / weight stream: measured losses, profiles/r03/gemm_cache_policy_ab.txt), "gemm256_gm" (tile-group size),
 * "gemm_fp8_sched" = 3 (default) | 0 (round 1's four-phase loop), "attn_kernel" = 7 (attn7.hip, default) | 2 (attn2.hip);
 * attention families 1, 3..6, 9 and "gemm256" = 3 | 4 (the two 4-wave GEMMs) are the measured-slower experiments under
 * csrc/experiments/, present only in a library built with ICV_EXPERIMENTS=1 ("require_experiments" returns 0 exactly then),
 * "attn<N>_variant" (bit flags, see each file; "attn7_variant": default 132 = 128-key publish + s_setprio, negative = default),
 * "attn7_short" (largest key count that takes attn7's 4-wave / two-stage launch shape: default 1024 = the cross-attention calls,
 * 0 = never, negative = default), "attn_defer_max_log2" (rescale threshold, default 8), "attn_unit_scale" = 0 | 1,
 * "ln_waves_per_row" = 0 | 1 | 2 | 4.  TIMING ABLATIONS that make results WRONG on purpose (tools/ only): "gemm256_ablate",
 * "gemm256x_ablate", "attn7_ablate".  The Python host also reads options from the environment:
 * ICV_OPTIONS="name=value,name=value". */
int icv_set_option(const char* name, int value);

/* ---- GEMM with fused epilogues (K1, K2, K4, K7, K9, K10, K11 of SURVEY §8a-3) ------------
 * C[M,N] = A[M,K] (bf16, lda) x W[N,K]^T (bf16, torch Linear layout, ldw)  (+ bias f32[N])
 * Replaces the nn.Linear / Conv3d-as-GEMM calls inside diffsynth's WanModel reached from
 * `self.pipe(...)` [R infinicube/videogen/inference.py:216-226].
 * K must be a multiple of 64; N a multiple of 4; M arbitrary.
 * Output column n lands in sub-buffer n / nsplit at column n % nsplit:
 *   addr = out + (n / nsplit) * split_stride + m * ldo + (n % nsplit)   (nsplit = N: plain)
 */
enum {
  ICV_EPI_BF16 = 0,       /* out bf16 = acc + bias                                        */
  ICV_EPI_GELU_BF16 = 1,  /* out bf16 = gelu_tanh(acc + bias)                   (K10)     */
  ICV_EPI_RESID_F32 = 2,  /* out f32  = resid[m,n] + gate[n] * (acc + bias)     (K7/K10)  */
  ICV_EPI_F32 = 3         /* out f32  = acc + bias                              (K11)     */
};
int icv_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                  int64_t M, int64_t N, int64_t K, int epilogue,
                  void* out, int64_t ldo, int64_t nsplit, int64_t split_stride,
                  const float* resid, int64_t ldr, const float* gate, void* stream);

/* ---- small-M fp32 GEMV (K2: time embedding / time projection) ----------------------------
 * out[m,n] = out_act( sum_k in_act(x[m,k]) * W[n,k] + bias[n] ),  x,out,bias f32; W bf16.
 * act: 0 = identity, 1 = SiLU.  M <= 8.  K multiple of 8. */
int icv_gemv_f32(const float* x, const void* W, const float* bias, float* out,
                 int64_t M, int64_t N, int64_t K, int in_act, int out_act, void* stream);

/* ---- K2 sinusoidal timestep embedding: out f32[dim] = cat[cos(t f_i), sin(t f_i)] -------- */
int icv_sinusoidal_embedding(double timestep, int64_t dim, float* out, void* stream);

/* ---- out[r, :] = a[r, :] + b[:]   (rows x n, f32): modulation + t_mod (Appendix A.4) ------ */
int icv_bcast_add_f32(const float* a, const float* b, float* out, int64_t rows, int64_t n,
                      void* stream);

/* ---- K3 / K8: LayerNorm (fp32 stats) [+ affine] [+ adaLN modulate] -> bf16 ----------------
 * out = (LN(x) * weight + bias) * (1 + scale) + shift ; any of weight/bias/scale/shift may be
 * NULL.  x f32 [rows, d] (ldx), out bf16 [rows, d] (ldo).  d multiple of 256, d <= 8192. */
int icv_ln_modulate(const float* x, int64_t ldx, const float* weight, const float* bias,
                    const float* shift, const float* scale, void* out, int64_t ldo,
                    int64_t rows, int64_t d, float eps, void* stream);

/* ---- fp8 path (BASELINE.json config #5: "fp8 MFMA weights ... CDNA4 fp8 path") ---------------
 * The reference names the I2V-14B checkpoint [R infinicube/videogen/download_checkpoint.py:24-29]; running its
 * projections in fp8 is this build's option (WanDiT(gemm_dtype="fp8")), not a behaviour of the reference.
 * Format: OCP e4m3 bytes + ONE f32 scale per row (token row of an activation, output-channel row of a
 * weight): x[r,k] ~= q[r,k] * scale[r], scale[r] = max_k|x[r,k]| / 448 (1 for an all-zero row), RNE.
 *
 * icv_quantize_rows_fp8: src bf16 (src_is_f32 = 0) or f32 [rows, K] (ld elements) -> out e4m3 [rows, K]
 *   (ldo bytes), scale f32 [rows].  K % 8 == 0.
 * icv_ln_modulate_fp8: icv_ln_modulate whose output row is quantised in the same kernel.
 * icv_gemm_fp8: out = epilogue((A . W^T) * a_scale[m] * w_scale[n] + bias), A e4m3 [M,K] (lda bytes),
 *   W e4m3 [N,K] (ldw bytes), K % 128 == 0; epilogues and split output exactly as icv_gemm_bf16.
 *   v_mfma_f32_16x16x128_f8f6f4: twice the bf16 MFMA rate. */
int icv_quantize_rows_fp8(const void* src, int src_is_f32, int64_t ld, int64_t rows, int64_t K,
                          void* out, int64_t ldo, float* scale, void* stream);
int icv_ln_modulate_fp8(const float* x, int64_t ldx, const float* weight, const float* bias,
                        const float* shift, const float* scale, void* out_fp8, int64_t ldo,
                        float* out_scale, int64_t rows, int64_t d, float eps, void* stream);
int icv_gemm_fp8(const void* A, int64_t lda, const float* a_scale, const void* W, int64_t ldw,
                 const float* w_scale, const float* bias, int64_t M, int64_t N, int64_t K, int epilogue,
                 void* out, int64_t ldo, int64_t nsplit, int64_t split_stride, const float* resid,
                 int64_t ldr, const float* gate, void* stream);

/* ---- K5: RMSNorm over the full model dim (+ 3-D RoPE), in place on bf16 -------------------
 * Up to two tensors per launch (q and k): x0/x1 bf16 [rows, d] (ld), w0/w1 f32 [d]; x1 may be
 * NULL.  RoPE is applied when rope_tab != NULL: rope_tab = f32 (cos,sin) pairs laid out
 * [T][22] ++ [Hp][21] ++ [Wp][21] (per-axis tables, angles computed in fp64 on the host),
 * token index of local row r = tok0 + r, token -> (f, h, w) with w fastest. */
int icv_rmsnorm_rope(void* x0, const float* w0, void* x1, const float* w1, int64_t ld,
                     int64_t rows, int64_t d, float eps, const float* rope_tab,
                     int64_t T, int64_t Hp, int64_t Wp, int64_t tok0, void* stream);

/* ---- K6 / K9: flash attention forward, non-causal, head_dim 128, bf16 in/out --------------
 * q [Sq, H*128] (ldq), k/v [Skv, H*128] (ldk/ldv), o [Sq, H*128] (ldo); fp32 softmax/accum.
 * softmax(q k^T * scale) v per head.  Sq, Skv arbitrary (>0). */
int icv_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                      int64_t ldv, void* o, int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads,
                      float scale, void* stream);

/* ---- K9, i2v branch: same as icv_attention_fwd but o += softmax(q k^T * scale) v (bf16 read-add-
 * write).  Wan2.1 i2v sums the cross-attention over the 257 CLIP image tokens with the one over the
 * text tokens (BASELINE.json config #5; [R infinicube/videogen/download_checkpoint.py:24-29] lists
 * the I2V-14B DiT and its CLIP encoder). */
int icv_attention_fwd_add(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                          int64_t ldv, void* o, int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads,
                          float scale, void* stream);

/* ---- K6 / K9 in the fp8 mode (BASELINE.json config #5, "CDNA4 fp8 path"): e4m3 attention ------------------
 * Both GEMMs of attention on v_mfma_scale_f32_32x32x64_f8f6f4.  Not a behaviour of the reference (it runs bf16
 * flash attention through diffsynth); selected by this build's WanDiT(attn_dtype="fp8").
 * icv_attention_fp8_prepare: q (may be NULL: keys / values only), k, v bf16 [S, H*128] -> per-head abs-max
 *   amax f32 [3, H] (q, k, v rows; power-of-two scales 2^ceil(log2(amax/448)) are derived from it), qq / kq e4m3
 *   [S, H*128] (ld in bytes), vt e4m3 transposed key-permuted tiles [H][ceil(Skv/64)][128][64]
 *   (icv_attention_fp8_vt_bytes gives its size).  K must already carry the softmax scale * log2(e) ("unit scale").
 *   q == NULL: keys / values only; k == NULL: queries only (the other rows of amax are left untouched).
 * icv_attention_fp8_fwd: o bf16 [Sq, H*128] = softmax2(qq kq^T) v  (exp2, i.e. natural softmax of the unscaled
 *   product when K carries (1/sqrt d) log2 e).
 * icv_attention_fp8_fwd_chunk: the same over one chunk of keys with the carried state of icv_attention_fwd_chunk
 *   (each chunk is prepared on its own: its own K / V scales; the state is in real units). */
int64_t icv_attention_fp8_vt_bytes(int64_t Skv, int64_t heads);
int icv_attention_fp8_prepare(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                              int64_t Sq, int64_t Skv, int64_t heads, void* qq, int64_t ldqq, void* kq,
                              int64_t ldkq, void* vt, float* amax, void* stream);
int icv_attention_fp8_fwd(const void* qq, int64_t ldqq, const void* kq, int64_t ldkq, const void* vt,
                          const float* amax, void* o, int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads,
                          void* stream);
int icv_attention_fp8_fwd_chunk(const void* qq, int64_t ldqq, const void* kq, int64_t ldkq, const void* vt,
                                const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml,
                                int64_t Sq, int64_t Skv, int64_t heads, int first, int last, void* stream);
/* e4m3 K|V ON THE WIRE (sequence parallel; config #5): every rank quantises its own K|V rows once and the exchange (K13) moves
 * e4m3 bytes — half the xGMI traffic of bf16 rows, 1/world of the quantise work of "gather, then quantise on every rank".
 * icv_attention_fp8_kv_amax: amax[1][h], amax[2][h] <- per-head abs-max of this rank's k / v rows (bf16 [rows, H*128]; row 0 of
 *   amax, the queries', is left alone).  The host max-reduces these 2*H floats over the ranks: every rank then uses the scale
 *   of the UNSHARDED launch, so the e4m3 values are the single-GPU ones.
 * icv_attention_fp8_quantize_kv: rows of k / v -> one "blob" = [ kq e4m3 [rows_pad, H*128] | vt tiles [H][rows_pad/64][128][64] ],
 *   rows_pad = rows rounded up to 64, icv_attention_fp8_blob_bytes(rows, H) bytes, with the scales derived from amax.
 * icv_attention_fp8_fwd_pieces: icv_attention_fp8_fwd_chunk over `n_pieces` such blobs back to back (the gathered chunk,
 *   rank-major), piece_rows real keys in each; padding keys of a piece's last tile are masked. */
int64_t icv_attention_fp8_blob_bytes(int64_t rows, int64_t heads);
int icv_attention_fp8_kv_amax(const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t rows, int64_t heads, float* amax,
                              void* stream);
int icv_attention_fp8_quantize_kv(const void* k, int64_t ldk, const void* v, int64_t ldv, int64_t rows, int64_t heads,
                                  const float* amax, void* blob, void* stream);
int icv_attention_fp8_fwd_pieces(const void* qq, int64_t ldqq, const void* blobs, int64_t piece_rows, int64_t n_pieces,
                                 const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int64_t Sq,
                                 int64_t heads, int first, int last, void* stream);
/* icv_attention_fp8_fwd_pieces over blobs that may still be ARRIVING (round 6; SURVEY.md §8e for the e4m3 wire format; the counterpart
 * of icv_attention_fwd_pieces below): the blobs are walked in the order seq_piece[0 .. n_pieces) - a permutation of the pieces, this
 * rank's own first - and position i is read once (int)(flags[seq_flag[i]] - seq_value[i]) >= 0 (seq_flag[i] < 0: there when the launch
 * starts); the wait is inside the kernel, bounded by timeout_us (0 = for ever): a flag that does not come sets
 * *err = 0x80000000 | position (first one wins; err may be NULL) and the launch finishes on whatever the slot holds.  own_blob (may be
 * NULL): piece own_index is read THERE instead of from its slot in `blobs` (the rank's own blob is not copied).  seq_* are host
 * arrays of n_pieces (<= ICV_ATTN_MAX_PIECES) entries; flags / err are device words.
 * Replaces: the host-side wait for a chunk's whole exchange in front of icv_attention_fp8_fwd_pieces [EXT: the fork's gather-then-attend]. */
int icv_attention_fp8_fwd_pieces_gated(const void* qq, int64_t ldqq, const void* blobs, int64_t piece_rows, int64_t n_pieces,
                                       const void* own_blob, int64_t own_index, const int32_t* seq_piece, const int32_t* seq_flag,
                                       const uint32_t* seq_value, const uint32_t* flags, uint32_t* err, int64_t timeout_us,
                                       const float* amax, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int64_t Sq,
                                       int64_t heads, int first, int last, void* stream);

/* Diagnostics: buf = device u64 [capacity][4] (NULL = off).  While set, every work-group b < capacity of the following
 * icv_attention_fwd / _fwd_chunk launches (attn7 kernel) writes {start, end in 100 MHz s_memrealtime ticks, HW_ID, XCC_ID}
 * to buf[b]: the round structure of a launch and its work-group -> XCD placement (tools/attn_round_trace.py). */
int icv_attention_trace(void* buf, int64_t capacity);

/* ---- K6 split along the KEY axis (K13 overlap): attention over one chunk of keys with a carried
 * online-softmax state, so the sequence-parallel path can consume K/V chunks as the RCCL all-gather
 * delivers them.  State = acc f32 [Sq, H*128] (ldacc; un-normalised O) + ml f32 [Sq, H, 2] (running
 * max, row sum).  first != 0: start from the empty state (acc/ml not read).  last != 0: normalise and
 * write o (bf16); otherwise write the state back.  Key order across chunks is irrelevant. */
int icv_attention_fwd_chunk(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                            int64_t ldv, void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml,
                            int64_t Sq, int64_t Skv, int64_t heads, float scale, int first, int last,
                            void* stream);

/* ---- K6 for the sequence-parallel schedule, ARRIVAL-DRIVEN (SURVEY §8e "process K/V chunks in arrival order (own shard first)
 * with online-softmax merging"): ONE launch per layer over a list of K|V pieces instead of one icv_attention_fwd_chunk launch per
 * row chunk.  Replaces: the fork's / xDiT-USP's "all-gather K and V, then flash_attention" of a sequence-parallel DiT layer [EXT]
 * (the reference runs one dense sequence on one GPU, [R infinicube/inference/guidance_buffer_generation.py:759-766]).
 *   o bf16 [Sq, H*128] = softmax(q [k_0; k_1; ...]^T * scale) [v_0; v_1; ...]     (key order = piece order; irrelevant to the result
 *   up to fp32 summation order)
 * pieces: HOST array of n_pieces {k, v bf16 [rows, H*128] with row strides ldk / ldv, rows, flag, value}.  flag < 0: the rows are in
 * place when the launch starts (this rank's own rows: listed FIRST, read where the K|V projection wrote them).  flag >= 0: the rows are
 * there once (int32)(flags[flag] - value) >= 0, where `flags` is DEVICE-visible memory written behind the transfer (icv_ipc_arrival's
 * words, or icv_flag_write on the stream that waited for a collective); the rows must not be read by anybody between the launch and that
 * moment (they are not: nothing else consumes a gathered buffer).  A work-group that reaches a piece before its rows waits inside the
 * kernel (one lane polls, s_sleep between polls); after timeout_us (0 = for ever) it stores 0x80000000 | piece index into *err
 * (device-visible uint32, may be NULL; first time-out wins) and goes on with whatever bytes are there - a dead peer is an error the host
 * finds in *err, never a hung queue.  Empty pieces (rows == 0) are skipped; at most ICV_ATTN_MAX_PIECES non-empty ones.
 * trace (diagnostics, may be NULL): device u64 [heads * ceil(Sq/256)][n_pieces] <- s_memrealtime tick (100 MHz) at which that
 * work-group started that piece. */
#define ICV_ATTN_MAX_PIECES 64
typedef struct icv_kv_piece {
  const void* k;
  const void* v;
  int64_t rows;
  int32_t flag;
  uint32_t value;
} icv_kv_piece;
int icv_attention_fwd_pieces(const void* q, int64_t ldq, const icv_kv_piece* pieces, int64_t n_pieces, int64_t ldk, int64_t ldv,
                             void* o, int64_t ldo, int64_t Sq, int64_t heads, float scale, const uint32_t* flags, uint32_t* err,
                             int64_t timeout_us, void* trace, void* stream);
/* flags[index] <- value with a system-scope release, enqueued on `stream` (one thread): the arrival flag of a piece whose rows were
 * delivered by something `stream` has waited for (an RCCL collective, a copy); delay_us > 0 first holds the stream for that long
 * (tests: a late peer). */
int icv_flag_write(uint32_t* flags, int64_t index, uint32_t value, int64_t delay_us, void* stream);

/* ---- K1: im2col for Conv3d(k = s = (1,2,2)) on a [C,T,H8,W8] f32 latent -------------------
 * out bf16 [n_tok, C*4] (ldo), row = token tok0 + r (f, hp, wp; wp fastest),
 * col = c*4 + y*2 + z  (== conv weight [d, C, 1, 2, 2] flattened). */
int icv_patchify(const float* latent, int64_t C, int64_t T, int64_t H8, int64_t W8,
                 void* out, int64_t ldo, int64_t tok0, int64_t n_tok, void* stream);

/* ---- K11 tail + K12: unpatchify + CFG combine + Euler step, fused -------------------------
 * For local tokens r in [0, n_tok): v = hu + cfg_scale * (hc - hu) (hu may be NULL -> v = hc);
 * latent[c, f, 2hp+y, 2wp+z] += v[r, (y*2+z)*C + c] * dsigma.   hc/hu f32 [n_tok, 4*C] (ldh).
 * If vel_out != NULL the combined velocity is also scattered there (same layout as latent).
 * round_bf16 != 0 ("reference rounding"): every intermediate a torch_dtype=bf16 pipeline materialises — both model
 * outputs, (hc - hu), cfg * (.), hu + (.), v * dsigma and the updated latent — is rounded to bf16 (values stay in
 * f32 storage); 0 = exact f32 arithmetic (default). */
int icv_unpatchify_cfg_euler(float* latent, float* vel_out, const float* hc, const float* hu,
                             int64_t ldh, float cfg_scale, float dsigma, int64_t C, int64_t T,
                             int64_t H8, int64_t W8, int64_t tok0, int64_t n_tok, int round_bf16, void* stream);

/* ---- coordinate guidance buffer (producer of the hot path's input) ---------------
 * Replaces `generate_coordinate_buffer_from_memory_global_norm` [R infinicube/utils/buffer_utils.py:180-265]
 * (+ `unproject_depth_torch` [R infinicube/utils/depth_utils.py:402-466]).  depth f32 [N,H,W] on device
 * (0 = infinitely far); kinv_host9 = K^-1 row-major (HOST pointer, 9 floats); cam_to_cam0 f32 [N,16] on
 * device = pose_0^-1 pose_n row-major.  Three passes, the [N,H,W,3] point map is never stored:
 *   valid_mask:    mask[i] = depth != 0 && z_cam0 < 1e6                       (u8 [N*H*W])
 *   gather_points: out[j,:] = point of pixel pixel_index[j]                   (the <=100000-point quantile sample)
 *   normalize:     out = sky ? 1 : (clip((P - mins)/ranges*2-1, -1, 1)+1)/2  -> f32 [N,H,W,3] and/or
 *                  u8 [N,H,W,3] = trunc(out*255) (what the caller feeds WanVideoGenerator);
 *                  has_valid == 0 reproduces the reference's "no finite point" branch (P * 0.5). */
int icv_coord_valid_mask(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                         int64_t N, int64_t H, int64_t W, unsigned char* mask, void* stream);
int icv_coord_gather_points(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                            int64_t N, int64_t H, int64_t W, const int64_t* pixel_index, int64_t n,
                            float* out, void* stream);
int icv_coord_normalize(const float* depth, const float* kinv_host9, const float* cam_to_cam0,
                        int64_t N, int64_t H, int64_t W, const float* mins_host3,
                        const float* ranges_host3, int has_valid, float* out_f32,
                        unsigned char* out_u8, void* stream);

/* ---- semantic / instance colour buffer -----------------------------------------
 * icv_semantic_to_color replaces `semantic_to_color` [R infinicube/utils/semantic_utils.py:88-101]:
 *   semantics i32 [n] (class index), class_rgb_lut f32 [n_classes,3] on device (= PALETTE[MAPPING]);
 *   writes f32 [n,3] and/or u8 [n,3] = trunc(colour*255) (the caller's conversion).
 * icv_instance_overlay_u8 replaces `generate_rgb_semantic_buffer` [R infinicube/utils/semantic_utils.py:104-131]:
 *   out = instance > 0 ? instance_rgb_lut65536[instance] : semantics_rgb   (all u8 RGB; instance i32 [n],
 *   low 16 bits used, like the reference's uint16 cast). */
int icv_semantic_to_color(const int* semantics, int64_t n, const float* class_rgb_lut, int n_classes,
                          float* out_f32, unsigned char* out_u8, void* stream);
int icv_instance_overlay_u8(const unsigned char* semantics_rgb, const int* instance, int64_t n,
                            const unsigned char* instance_rgb_lut65536, unsigned char* out, void* stream);

/* ---- depth wire format ----------------------------------------------------------
 * icv_depth_to_u16 replaces `(depth_np * 100).astype(np.uint16)`, the payload of every
 * `NNNNNN.voxel_depth_100.front.png` member of `voxel_depth_100_<res>_front.tar`
 * [R infinicube/inference/guidance_buffer_generation.py:668-672]: out[i] = (uint16)(int64)(depth[i] * scale),
 * one rounded f32 multiply, truncation toward zero, wrap modulo 2^16.  depth f32 [n] (16-byte aligned), out u16 [n]. */
int icv_depth_to_u16(const float* depth, int64_t n, float scale, unsigned short* out, void* stream);

/* ---- the Wan-VAE's channel RMS norm (+ SiLU) as ONE pass over NDHWC rows ----------------------
 * Replaces, inside the tiled VAE encode / decode the reference asks for with `tiled=True`
 * [R infinicube/videogen/inference.py:69,171,225], the fork's `RMS_norm` (F.normalize over channels * sqrt(C) * gamma)
 * and the `nn.SiLU` that follows it in every residual block and head ([EXT] public Wan2.1 VAE): as stock ops five
 * elementwise / reduction passes over the activation.  x, out bf16 [rows, C] (C contiguous: the pixels of an NDHWC / NHWC
 * buffer; out may alias x), gamma f32 [C]:   y = x / max(|x|_2, eps) * scale * gamma;  act = 1: y = y * sigmoid(y).
 * fp32 inside, one rounding to bf16.  C % 8 == 0, C <= 2048. */
int icv_rmsnorm_act_rows(const void* x, void* out, const float* gamma, int64_t rows, int64_t C, float scale,
                         float eps, int act, void* stream);
/* The same norm (+ SiLU) over a PADDED NDHWC volume [Tp, Hp, Wp, C] (csrc/conv.hip's activation layout: `pt` leading padding
 * frames, a one-pixel spatial halo): interior positions as icv_rmsnorm_act_rows, padding / halo rows are WRITTEN AS ZEROS
 * whatever the input holds there — the output is a valid zero-padded input of the next icv_conv3d_ndhwc. */
int icv_rmsnorm_act_volume(const void* x, void* out, const float* gamma, int64_t Tp, int64_t Hp, int64_t Wp, int64_t pt, int64_t C,
                           float scale, float eps, int act, void* stream);

/* ---- voxel ray-cast of the guidance-buffer renderer ------------------------------------------
 * Replaces the three fVDB-bound calls of `generate_infinicube_buffer_from_fvdb_grid`
 * [R infinicube/utils/fvdb_utils.py:572-605]: `get_zdepth_map_from_voxel` (`segments_along_rays(o, d, 1, eps=1e-1)`
 * [R infinicube/camera/base.py:520-571]) and `get_semantic_map_from_voxel` x 2 (`voxels_along_rays(o, d, 1, eps=1e-2)`
 * [R infinicube/camera/base.py:573-619]) on a grid built by `points_to_fvdb` [R infinicube/utils/fvdb_utils.py:71-215].
 * The voxel world is a DENSE int32 index volume in HBM (vol[z][y][x] = voxel index or -1; dims multiples of 8) plus one
 * occupancy byte per 8^3 brick.
 * icv_voxel_scatter: ijk i32 [M,3] (unique occupied voxels) -> vol / bricks (pre-filled with -1 / 0); vol_min3 = ijk of
 *   cell (0,0,0) of the volume (host ints), dims3 = (Dx, Dy, Dz).
 * icv_voxel_raycast: grid_lo3 = world coordinate of the low corner of cell (0,0,0), voxel_size3 (host floats);
 *   rays_cam f32 [HW,3] = the camera model's normalised rays, poses f32 [N,16] camera-to-world (device);
 *   depth_out f32 [N,HW] = z-depth of the first occupied run longer than eps_depth (0 = none);
 *   attrK_out i32 [N,HW] = attrK[voxel] of the first occupied voxel crossed for more than eps_voxel, else backgroundK;
 *   index_out = that voxel's index or -1.  Outputs may be NULL. */
int icv_voxel_scatter(const int* ijk, int64_t M, const int* vol_min3, const int* dims3, int* vol,
                      unsigned char* bricks, void* stream);
int icv_voxel_raycast(const int* vol, const unsigned char* bricks, const int* dims3, const float* grid_lo3,
                      const float* voxel_size3, const float* rays_cam, const float* poses, int64_t N,
                      int64_t HW, float eps_depth, float eps_voxel, const int* attr0, const int* attr1,
                      int background0, int background1, float* depth_out, int* attr0_out, int* attr1_out,
                      int* index_out, void* stream);

/* ---- context-style driver: the whole DiT forward of a token shard in ONE call (SURVEY §8b B-native) ----------------
 * Replaces one `WanModel.forward` of the diffsynth fork reached from `self.pipe(...)`
 * [R infinicube/videogen/inference.py:216-226] for a host that does not drive the per-op entry points itself; it calls
 * exactly those launchers in the order infinicube_amd/videogen/dit.py does (bit-identical results), in every mode of that
 * driver: bf16 or e4m3 projections / self-attention, t2v or i2v, one rank or the sequence-parallel schedule.
 * Every tensor is BORROWED: `icv_dit_bind` records a device pointer, nothing is copied or owned.
 *   per-layer names (layer >= 0): wqkv [3d,d] bf16, bqkv f32 [3d], nq / nk f32 [d] (nk carries the folded softmax scale),
 *     wo, bo, n3w, n3b, xq_w, xq_b, xnq, xo_w, xo_b, f0_w [ffn,d], f0_b, f2_w [d,ffn], f2_b;
 *     e4m3 projections (config #5): bind the e4m3 rows under the weight's name AND its f32 per-output-row scales under
 *     "<name>_s" (wqkv_s, wo_s, xq_w_s, xo_w_s, f0_w_s, f2_w_s) — a weight with scales is taken as e4m3, any subset may be;
 *   global names (layer = -1): patch_w bf16 [d,k_patch], patch_b, head_w bf16 [out_cols,d], head_b, rope (f32 table of
 *     ops.RopeTable), workspace x f32 [n,d], x_stem f32 [n,d] (optional), h bf16 [n,d], qkv bf16 [3,n,d], att bf16 [n,d],
 *     ff bf16 [n,ffn], patches bf16 [n,k_patch];
 *     e4m3 projections: h8 / att8 e4m3 [n,d], ff8 e4m3 [n,ffn] and their row scales h8s / att8s / ff8s f32 [n];
 *     e4m3 self-attention (icv_dit_set_fp8): a8_qq e4m3 [n,d], a8_kq e4m3 [kv_rows,d], a8_vt (icv_attention_fp8_vt_bytes),
 *     a8_amax f32 [3,heads];
 *     sequence-parallel (icv_dit_set_seqpar): kv_loc bf16 [n,2d] (row = k | v), kv_full bf16 [world*n,2d] (chunk-major,
 *     rank-major inside), sp_acc f32 [n,d], sp_ml f32 [n,heads,2].
 * icv_dit_set_fp8: attn_fp8 != 0 routes self-attention through the e4m3 kernels (cross-attention stays bf16).
 * icv_dit_set_seqpar: this context's shard is one of `world` token shards; per layer the K|V rows [bounds[c], bounds[c+1])
 *   of every rank are exchanged with icv_allgather_kv on `comm`, enqueued on `side_stream` (a hipStream_t the caller owns)
 *   and fenced against the launch stream with events the context owns, and attention consumes the chunks in order with
 *   carried softmax state (icv_attention_fwd_chunk).  comm == NULL returns to the single-rank schedule.
 *   SCOPE: the one-call driver knows THIS sequence-parallel form only - an RCCL communicator (icv_comm_*), bf16 rows on the wire,
 *   one carried-state launch per row chunk.  The copy-engine transport (icv_ipc_*), e4m3 K|V on the wire
 *   (icv_attention_fp8_quantize_kv / _fwd_pieces) and the arrival-driven attention (icv_attention_fwd_pieces) are driven through
 *   the per-op entry points by the host (infinicube_amd/videogen/dit.py, which refuses the one-call driver in those modes).
 * icv_dit_forward: latent f32 [C,T,H8,W8]; mod f32 [layers,6d] and hmod f32 [2,d] = this step's modulation tables;
 *   ctx_k / ctx_v bf16 [ctx_len, d] of layer 0, layer i at + i * ctx_layer_stride elements (text K/V cache); img_k / img_v
 *   likewise with img_len / img_layer_stride, or NULL; buf_tokens f32
 *   [n,d] or NULL; head_out f32 [n,out_cols]; num_layers < 0 = all; stem 0 | 1 = save x after layer 0's self-attention
 *   block into x_stem | 2 = start from x_stem (the context-free stem shared by the two CFG forwards). */
typedef struct icv_dit icv_dit;
typedef struct {
  int64_t dim, ffn_dim, heads, layers;
  int64_t n_tok, tok0;       /* this shard's tokens [tok0, tok0 + n_tok) of the (T, Hp, Wp) grid */
  int64_t T, Hp, Wp;
  int64_t k_patch, out_cols; /* padded patch-GEMM K (multiple of 64); head columns = out_dim * 4 */
  float eps;
} icv_dit_config;
int icv_dit_create(const icv_dit_config* cfg, icv_dit** out);
void icv_dit_destroy(icv_dit* ctx);
int icv_dit_bind(icv_dit* ctx, const char* name, int64_t layer, const void* device_ptr);
typedef struct icv_comm icv_comm;
int icv_dit_set_fp8(icv_dit* ctx, int attn_fp8);
int icv_dit_set_seqpar(icv_dit* ctx, icv_comm* comm, int64_t world, int64_t n_chunks, const int64_t* bounds, void* side_stream);
int icv_dit_forward(icv_dit* ctx, const float* latent, int64_t C, int64_t H8, int64_t W8, const float* mod,
                    const float* hmod, const void* ctx_k, const void* ctx_v, int64_t ctx_len,
                    int64_t ctx_layer_stride, const void* img_k, const void* img_v, int64_t img_len,
                    int64_t img_layer_stride, const float* buf_tokens, float* head_out, int64_t num_layers, int stem,
                    float attn_scale, void* stream);

/* Per-launch timing of the dominant kernel (self-attention, K6) inside icv_dit_forward: with profiling enabled every
 * forward records a HIP event pair around that launch ON THE LAUNCH STREAM; icv_dit_profile_read waits for the recorded
 * events, returns their summed duration and count, and resets the list.  bench.py's roofline figure uses it.  Under
 * icv_dit_set_seqpar one pair is recorded per KEY-CHUNK launch (after the stream's wait on that chunk's transfer), so
 * `launches` then counts chunk launches: chunks x layers per forward.  Leave it off under graph capture. */
int icv_dit_profile(icv_dit* ctx, int enable);
int icv_dit_profile_read(icv_dit* ctx, double* total_ms, int64_t* launches);

/* ---- the Wan-VAE's convolutions as shifted-row GEMMs on the matrix cores (csrc/conv.hip) ---------
 * Replaces, inside the tiled VAE encode / decode [R infinicube/videogen/inference.py:69,171,225], every nn.Conv3d / nn.Conv2d
 * of the fork's Wan-VAE ([EXT] public Wan2.1 VAE: causal 3x3x3, (3,1,1), per-frame 3x3, 1x1x1) — stock PyTorch hands these
 * to MIOpen, whose first call per shape searches kernels for tens of seconds and whose choice differs between processes.
 * The activation is a PADDED NDHWC volume flattened to rows: x bf16 [rows, ldx] where the causal / spatial zero padding
 * are real rows; a tap (dt, dh, dw) is ONE row offset (dt*Hp + dh)*Wp + dw for the whole volume:
 *     out[m, 0:cout] = bias + sum_i  x[m + tap_row_offsets[i], 0:cin] . w[:, i*cin : (i+1)*cin]^T  (+ resid[m, 0:cout])
 * for every m in [m0, m1) — halo rows included (their results are garbage by construction; the consumer masks them).
 * w bf16 [cout, K], K = ntaps*cin rounded up to a multiple of 64 with zeros, K-contiguous, tap-major / channel-minor;
 * cin % 32 == 0, cout % 4 == 0 (pad with zero channels / filters); out / resid bf16 with row strides ldo / ldr.
 * x points at row 0; x_rows_before / x_rows_after = how many addressable rows the buffer has before row 0 and after row
 * m1 - 1 (every m + offset must fall inside; checked).  fp32 accumulate, one rounding to bf16. */
int icv_conv3d_ndhwc(const void* x, int64_t ldx, int64_t x_rows_before, int64_t x_rows_after, const void* w, const float* bias,
                     const int64_t* tap_row_offsets, int64_t ntaps, int64_t cin, int64_t m0, int64_t m1, int64_t cout, void* out,
                     int64_t ldo, const void* resid, int64_t ldr, void* stream);

/* ---- K13: the sequence-parallel K|V exchange as a C entry point ---------------------------------
 * Replaces: the fork's sequence-parallel attention gather (north_star "per-layer K/V all-gather"; upstream xDiT/USP
 * all-gather of K and V inside `flash_attention`), for hosts that do not use torch.distributed (which
 * infinicube_amd/videogen/seqpar.py does by default).  RCCL is dlopen'ed at first use.  rank 0 of the group makes the id,
 * the host ships its ICV_COMM_ID_BYTES to the other ranks, every rank creates its communicator on its CURRENT device.
 * icv_allgather_kv: rows [m, row_bytes] of every rank -> out [world * m, row_bytes], rank-major, enqueued on `stream`. */
#define ICV_COMM_ID_BYTES 128
int icv_comm_unique_id(char* id);
int icv_comm_create(const char* id, int rank, int world, icv_comm** out);
void icv_comm_destroy(icv_comm* comm);
int icv_allgather_kv(icv_comm* comm, const void* rows, void* out, int64_t m, int64_t row_bytes, void* stream);

/* ---- K13 without compute units: copy-engine pulls of the peers' K|V rows (csrc/ipc.hip) -------------
 * Replaces: the same per-layer K|V exchange as icv_allgather_kv (north_star "all-gather of K/V over xGMI"), for the case
 * where the RCCL channel kernels cost the overlapped attention more than the transfer is worth (one attention work-group
 * per CU: a CU taken by a channel stretches a whole round — profiles/r05/kv_contention.md).  Every rank owns a SYMMETRIC
 * HEAP (same size everywhere) that holds its K|V rows; peers open it through hipIpc and PULL row chunks with
 * hipMemcpyAsync on one stream per peer (SDMA over the pair's xGMI link); readiness and reuse are flag words in a POSIX
 * shared-memory segment `shm_name` (mapped and hipHostRegister'ed by every rank) written with hipStreamWriteValue32 and
 * waited for by one-wave kernels with a deadline — no host round trip anywhere in the per-layer path (on current ROCm the
 * runtime's own hipStreamWaitValue32 is such a spinning kernel too, without a deadline; the wait is resident for the skew
 * between two ranks; the ROWS move by SDMA).
 *   every rank:  icv_ipc_create(name, rank, world, heap, heap_bytes, &ipc) — `heap` = a BORROWED device buffer of heap_bytes
 *                (it must stay alive until icv_ipc_destroy; the allocation containing it is what gets exported), or NULL
 *                to let the library hipMalloc one (icv_ipc_heap returns it); icv_ipc_export(ipc, handle) -> host ships the
 *                ICV_IPC_HANDLE_BYTES of every rank to every rank (any side channel); icv_ipc_open_peer(ipc, p, handle_p);
 *                after everyone has mapped the segment one rank may icv_ipc_shm_unlink(name) (nothing is left in /dev/shm).
 *   per layer:   icv_ipc_acquire(ipc, stream)  BEFORE the kernels that overwrite heap rows (waits until every peer has
 *                pulled everything published so far);  per row chunk  icv_ipc_gather_start(ipc, src_offset, bytes, out,
 *                stream, &ticket): the `bytes` at `src_offset` of EVERY rank's heap -> out [world * bytes], rank-major;
 *                the rows must have been produced on `stream`.  icv_ipc_gather_wait(ipc, ticket, stream): `stream` waits
 *                until that chunk has landed.  Every rank must issue the same sequence of gather_start calls; at most
 *                ICV_IPC_SLOTS exchanges may be un-waited at a time.
 * Needs hipDeviceAttributeCanUseStreamWaitValue and working hipIpc between the ranks' devices: icv_ipc_create /
 * icv_ipc_open_peer fail (non-zero, text in icv_last_error) where either is refused and the host drops the transport. */
#define ICV_IPC_HANDLE_BYTES 72 /* hipIpcMemHandle_t of the allocation + the heap's byte offset inside it */
#define ICV_IPC_SLOTS 32
typedef struct icv_ipc icv_ipc;
int icv_ipc_create(const char* shm_name, int rank, int world, void* heap, int64_t heap_bytes, icv_ipc** out);
void icv_ipc_destroy(icv_ipc* ipc);
int icv_ipc_shm_unlink(const char* shm_name);
int icv_ipc_heap(icv_ipc* ipc, void** base, int64_t* bytes);
int icv_ipc_export(icv_ipc* ipc, char* handle);
int icv_ipc_open_peer(icv_ipc* ipc, int peer, const char* handle);
int icv_ipc_gather_start(icv_ipc* ipc, int64_t src_offset, int64_t bytes, void* out, void* stream, int64_t* ticket);
int icv_ipc_gather_wait(icv_ipc* ipc, int64_t ticket, void* stream);
int icv_ipc_acquire(icv_ipc* ipc, void* stream);
int64_t icv_ipc_tickets(const icv_ipc* ipc);
/* Arrival flags for icv_attention_fwd_pieces: *flags = DEVICE memory uint32 [world]; flags[p] >= t + 1 once the rows rank p
 * contributed to ticket t have landed in that ticket's `out` (monotonic; entry [own rank] is never written: own rows are read in place).
 * A consumer that gates on these calls icv_ipc_gather_consumed(ipc, ticket) instead of icv_ipc_gather_wait (bookkeeping only: no
 * stream waits are enqueued).
 * icv_ipc_configure(ipc, copy_own_rows): 0 = gather_start no longer copies this rank's own rows into `out` (the arrival-driven
 * attention reads them from the heap). */
int icv_ipc_arrival(icv_ipc* ipc, const uint32_t** flags);
int icv_ipc_gather_consumed(icv_ipc* ipc, int64_t ticket);
int icv_ipc_configure(icv_ipc* ipc, int copy_own_rows);
/* Liveness (round 6).  Every device-side wait of the transport is a one-wave kernel with a deadline (ICV_IPC_WAIT_TIMEOUT_MS, default
 * 60000, 0 = none): a wait that expires records whom it was waiting for and lets its queue go on with stale rows.
 * icv_ipc_check: 0 = no wait of this rank has expired; otherwise non-zero and icv_last_error names the peer - the run is invalid from
 * that exchange on (call it once per denoising step; it reads one host word).
 * icv_ipc_drain(ipc, timeout_ms): give this rank's queues that long to finish on their own, then satisfy every wait word of the segment
 * so that they do (0 = drained by themselves, 1 = released): teardown never depends on a live peer.  icv_ipc_destroy calls it
 * (ICV_IPC_DRAIN_TIMEOUT_MS, default 5000). */
int icv_ipc_check(icv_ipc* ipc);
int icv_ipc_drain(icv_ipc* ipc, int timeout_ms);
/* First-contact probe: is a pull of `bytes` from rank `peer`'s heap executed WITHOUT compute units?  An occupier kernel takes every
 * wave slot of this device for ~8 ms; meanwhile a control kernel and the copy are enqueued on other streams.  *kind: 1 = the copy
 * finished while no wave slot was free (a copy engine moved it), 2 = it did not (the runtime used a blit kernel: the transport then
 * costs the overlapped attention CUs like an RCCL channel does), 0 = inconclusive (the control kernel got a slot).  *copy_ms
 * (may be NULL): the copy's duration from events.  A diagnostic: ~20 ms, synchronises the device, not for the per-layer path. */
int icv_ipc_probe_copy(icv_ipc* ipc, int peer, int64_t bytes, int* kind, double* copy_ms);
/* The same probe for any copy hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault) can make (device, peer-device, pinned host pointers;
 * dst == NULL = a scratch device buffer): the instrument's own controls - a same-device copy must answer 2, a pinned-host-to-device
 * copy 1 (tests/test_kernels_gpu.py). */
int icv_probe_copy_path(const void* src, void* dst, int64_t bytes, int* kind, double* copy_ms);
/* a rank that cannot go on releases every peer wait that depends on it (its flag words jump past every sequence number: the
 * peers pull undefined bytes instead of spinning forever) and refuses further exchanges; the error itself travels by the host's
 * own channel.  Turns "one rank failed" from a hang on the others into an error on all. */
int icv_ipc_abort(icv_ipc* ipc);

/* ---- dtype plumbing: f32 -> bf16 (round-to-nearest-even), n elements ---------------------- */
int icv_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICVIDEO_H_ */
