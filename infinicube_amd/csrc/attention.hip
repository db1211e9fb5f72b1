// Attention entry points of the C ABI (include/icvideo.h) and the kernel-family routing.
//   7 = attn7.hip (default: LDS-DMA ring + lazy max + unit scale), 2 = attn2.hip (the previous default, register-staged
//   128-key tile).  Families 1, 3, 4, 5, 6, 9 are measured-slower (or tying) experiments kept for A/B under experiments/; they are
//   compiled in only when the library is built with ICV_EXPERIMENTS=1 (csrc/build.sh) and otherwise report an error.
#include "icv_common.h"


int icv_attn1_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                       int64_t ldo, int64_t Sq, int64_t Skv, int64_t heads, float scale, void* stream);
int icv_attn2_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);

int icv_attn3_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);

int icv_attn9_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);
int icv_attn7p_single(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* acc,
                      int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, hipStream_t st);
int icv_attn7q_single(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* acc,
                      int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int variant,
                      hipStream_t st);
int icv_attn7_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);
int icv_attn6_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);
int icv_attn5_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);
int icv_attn4_dispatch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in,
                       int state_out, int64_t Sq, int64_t Skv, int64_t heads, float scale, int var,
                       hipStream_t st);

// Kernel family selection (icv_set_option("attn_kernel", n)): 7 = attn7.hip (default: LDS-DMA ring + lazy max + unit
// scale), 2 = attn2.hip, 3..6 = the experiments kept for A/B, 1 = experiments/attn1.hip (icv_attention_fwd only).
constexpr int ATTN_KERNEL_DEFAULT = 7;
constexpr int ATTN7_VARIANT_DEFAULT = 132;
constexpr int ATTN7Q_DEFAULT = 0;
static int attn_route(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                      int64_t ldo, float* acc, int64_t ldacc, float* ml, int state_in, int state_out, int64_t Sq,
                      int64_t Skv, int64_t heads, float scale, hipStream_t st) {
#define ATT_ARGS q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, state_in, state_out, Sq, Skv, heads, scale
  switch (icv_get_option_int("attn_kernel", ATTN_KERNEL_DEFAULT)) {
    case 7: {
      // default variant 132 = 128-key publish (one vmcnt(0) + barrier per TWO key tiles) + s_setprio around the MFMA
      // clusters: +1.4 % on the 37 440-key self-attention of both model sizes, neutral on the sequence-parallel shard
      // shapes (same-process A/B, profiles/r03/attention_variants.md); a negative option value = this default
      const int var7 = icv_get_option_int("attn7_variant", -1);
      // round 6: the default long-key launch (self-attention, plain or carried-state chunk) runs as ONE piece of the pieces kernel
      // (csrc/attn7p.hip: attn7's variant-132 schedule, bit-identical, 1.5-2.7 % faster at the 14B shapes); an explicit attn7_variant,
      // attn7_plain = 1 (A/B), the ablation switches and the short-key shape (cross-attention) keep attn7.hip
      int short_max = icv_get_option_int("attn7_short", -1);
      if (short_max < 0) short_max = 1024;
      if (var7 < 0 && Skv > short_max && !icv_get_option_int("attn7_plain", 0) && !icv_get_option_int("attn7_ablate", 0) && state_out != 2) {
#ifdef ICV_EXPERIMENTS
        // experiments/attn7q.hip: attn8's software-pipelined loop on the bf16 MFMA (1 / 2 = fragment prefetch distance).  Measured SLOWER than
        // attn7p (1133 vs 1196 TF/s): same MFMA busy (0.58 vs 0.59), 43 % more vector instructions, lower sustained clock (1.89 vs 1.94 GHz) -
        // the bf16 attention sits on the power limit, not on its schedule (profiles/r06/attn7q_pipelined_bf16_negative.txt)
        const int q7 = icv_get_option_int("attn7q", ATTN7Q_DEFAULT);
        if (q7 > 0) {
          const int rc = icv_attn7q_single(ATT_ARGS, q7, st);
          if (rc >= 0) return rc;
        }
#endif
        return icv_attn7p_single(ATT_ARGS, st);
      }
      return icv_attn7_dispatch(ATT_ARGS, var7 < 0 ? ATTN7_VARIANT_DEFAULT : var7, st);
    }
#ifdef ICV_EXPERIMENTS
    case 9: return icv_attn9_dispatch(ATT_ARGS, icv_get_option_int("attn9_variant", 0), st);
    case 6: return icv_attn6_dispatch(ATT_ARGS, icv_get_option_int("attn6_variant", 5), st);
    case 5: return icv_attn5_dispatch(ATT_ARGS, 0, st);
    case 4: return icv_attn4_dispatch(ATT_ARGS, icv_get_option_int("attn4_variant", 4), st);
    case 3: return icv_attn3_dispatch(ATT_ARGS, icv_get_option_int("attn3_variant", 0), st);
#else
    case 9: case 6: case 5: case 4: case 3:
      icv_set_error("attn_kernel 3..6 and 9 are experiments: rebuild libicvideo with ICV_EXPERIMENTS=1");
      return 1;
#endif
    default: return icv_attn2_dispatch(ATT_ARGS, icv_get_option_int("attn2_variant", 12), st);
  }
#undef ATT_ARGS
}

extern "C" int icv_attention_fwd_chunk(const void* q, int64_t ldq, const void* k, int64_t ldk,
                                       const void* v, int64_t ldv, void* o, int64_t ldo, float* acc,
                                       int64_t ldacc, float* ml, int64_t Sq, int64_t Skv,
                                       int64_t heads, float scale, int first, int last, void* stream) {
  ICV_REQUIRE(q && k && v, "icv_attention_fwd_chunk: null pointer");
  ICV_REQUIRE(Sq > 0 && Skv > 0 && heads > 0, "icv_attention_fwd_chunk: empty problem");
  ICV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "icv_attention_fwd_chunk: leading dims must keep 16-byte row alignment");
  ICV_REQUIRE((first && last) || (acc && ml && ldacc % 4 == 0), "icv_attention_fwd_chunk: carried state buffers required unless first && last");
  ICV_REQUIRE(!last || (o && ldo % 4 == 0), "icv_attention_fwd_chunk: output required for the last chunk");
  return attn_route(q, ldq, k, ldk, v, ldv, o, ldo, acc, ldacc, ml, first ? 0 : 1, last ? 0 : 1, Sq, Skv, heads, scale, (hipStream_t)stream);
}

extern "C" int icv_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk,
                                 const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                 int64_t Skv, int64_t heads, float scale, void* stream) {
  ICV_REQUIRE(q && k && v && o, "icv_attention_fwd: null pointer");
  ICV_REQUIRE(Sq > 0 && Skv > 0 && heads > 0, "icv_attention_fwd: empty problem (Sq=%lld Skv=%lld heads=%lld)", (long long)Sq, (long long)Skv, (long long)heads);
  ICV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "icv_attention_fwd: leading dims must keep 16-byte row alignment");
  if (icv_get_option_int("attn_kernel", ATTN_KERNEL_DEFAULT) != 1)
    return attn_route(q, ldq, k, ldk, v, ldv, o, ldo, nullptr, 0, nullptr, 0, 0, Sq, Skv, heads, scale, (hipStream_t)stream);
#ifdef ICV_EXPERIMENTS
  return icv_attn1_dispatch(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Skv, heads, scale, stream);
#else
  icv_set_error("attn_kernel 1 is an experiment: rebuild libicvideo with ICV_EXPERIMENTS=1");
  return 1;
#endif
}

extern "C" int icv_attention_fwd_add(const void* q, int64_t ldq, const void* k, int64_t ldk,
                                     const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                     int64_t Skv, int64_t heads, float scale, void* stream) {
  ICV_REQUIRE(q && k && v && o, "icv_attention_fwd_add: null pointer");
  ICV_REQUIRE(Sq > 0 && Skv > 0 && heads > 0, "icv_attention_fwd_add: empty problem (Sq=%lld Skv=%lld heads=%lld)", (long long)Sq, (long long)Skv, (long long)heads);
  ICV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "icv_attention_fwd_add: leading dims must keep 16-byte row alignment");
  return attn_route(q, ldq, k, ldk, v, ldv, o, ldo, nullptr, 0, nullptr, 0, 2, Sq, Skv, heads, scale, (hipStream_t)stream);
}

// ---- diagnostics: where and when every work-group of the NEXT attn7 launches runs -----------------------------------------
// icv_attention_trace(buf, capacity): buf = device u64 [capacity][4] (NULL switches tracing off).  While set, every attn7
// work-group b < capacity writes buf[b] = {start, end (s_memrealtime ticks, 100 MHz), HW_ID, XCC_ID} - the round structure and
// the work-group -> XCD placement of a launch (tools/attn_round_trace.py; profiles/r05/attn_round_occupancy.md).
namespace {
unsigned long long* g_trace = nullptr;
int g_trace_cap = 0;
}  // namespace
unsigned long long* icv_attention_trace_buffer(int* capacity) {
  *capacity = g_trace_cap;
  return g_trace;
}
extern "C" int icv_attention_trace(void* buf, int64_t capacity) {
  ICV_REQUIRE(capacity >= 0 && capacity < (1LL << 31), "icv_attention_trace: bad capacity");
  g_trace = (unsigned long long*)buf;
  g_trace_cap = buf ? (int)capacity : 0;
  return 0;
}
