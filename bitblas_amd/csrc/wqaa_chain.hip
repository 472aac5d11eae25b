// wqaa_chain.hip - wqaa_matmul_chain: a chain of DEPENDENT operators (the post-attention half of a decoder layer: o_proj
// (+ residual) -> RMSNorm -> gate / up * silu -> down_proj (+ residual); the reference's callers: integration/BitNet/
// modeling_bitnet.py:240-244, :839-860) described once and run as the launches it stands for, in order, every launch with the
// callers' elementwise ops folded in (wqaa_matmul_ex, wqaa_matmul_gate_up).
//
// Rounds 3-5 also carried a persistent one-launch member for these chains (a loader wave streaming the next operator's weights
// while consumers waited on 8-byte hand-off granules).  It was bit-identical and 50 % slower than the launches (37.6 vs 25.0 us
// per layer tail, profiles/r05_bench_members.json `chain_tail`; time line in profiles/r04_chain_lab.txt), needed every CU of the
// chip co-resident, and lies outside the operator boundary of SURVEY.md section 8: removed in round 6 (docs/DESIGN_r04.md
// section 3.3c keeps the post-mortem).  The entry point keeps its meaning: the launches.
#include "wqaa_common.h"

namespace wqaa {

// checks that make the chain ill-formed whatever runs it
static int chain_validate(const wqaa_chain_item* items, int count, int m) {
  if (!items || count < 1 || count > WQAA_CHAIN_MAX || m < 0) {
    set_error(WQAA_ERR_BAD_DESC, "matmul_chain: bad arguments (items=%p count=%d m=%d; at most %d items)", (const void*)items, count, m, WQAA_CHAIN_MAX);
    return WQAA_ERR_BAD_DESC;
  }
  for (int i = 0; i < count; ++i) {
    const wqaa_chain_item& it = items[i];
    if (!it.desc || it.desc->struct_size != (int32_t)sizeof(wqaa_matmul_desc) || it.desc->N <= 0 || it.desc->K <= 0) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d has a missing or malformed descriptor", i);
      return WQAA_ERR_BAD_DESC;
    }
    const wqaa_matmul_desc& d = *it.desc;
    if (!it.C) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d has no output buffer (every item is a launch of its own and stores its result)", i);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.kind != 0 && it.kind != 1) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d kind %d (0 matmul, 1 gate / up pair)", i, it.kind);
      return WQAA_ERR_BAD_DESC;
    }
    const bool pair = it.kind == 1;
    if (!it.B || (d.with_scaling && !it.Scale) || (d.zeros_mode != WQAA_Z_NONE && !it.Zeros) || (d.with_bias && !it.Bias) ||
        (pair && (!it.B2 || (d.with_scaling && !it.Scale2) || (d.zeros_mode != WQAA_Z_NONE && !it.Zeros2) || (d.with_bias && !it.Bias2)))) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d has a null operand its descriptor requires", i);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.input_from >= i || it.residual_from >= i || (it.input_from < 0 && !it.A)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d reads input %d / residual %d: an earlier item, or -1 with the pointer", i, it.input_from,
                it.residual_from);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.input_from >= 0 && items[it.input_from].desc->N != d.K) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d (K = %d) reads the output of item %d (N = %d)", i, d.K, it.input_from,
                items[it.input_from].desc->N);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.residual_from >= 0 && items[it.residual_from].desc->N != d.N) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d (N = %d) adds the output of item %d (N = %d)", i, d.N, it.residual_from,
                items[it.residual_from].desc->N);
      return WQAA_ERR_BAD_DESC;
    }
    if (pair && (it.residual || it.residual_from >= 0)) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d: a gate / up pair takes no residual", i);
      return WQAA_ERR_BAD_DESC;
    }
    if (it.residual && it.residual_from >= 0) {
      set_error(WQAA_ERR_BAD_DESC, "matmul_chain: item %d names two residuals", i);
      return WQAA_ERR_BAD_DESC;
    }
    if ((it.residual || it.residual_from >= 0) && it.norm_weight) {
      set_error(WQAA_ERR_UNSUPPORTED, "matmul_chain: item %d: RMSNorm input and residual add on one operator (wqaa_matmul_ex has no such launch)", i);
      return WQAA_ERR_UNSUPPORTED;
    }
  }
  return WQAA_OK;
}

// ---- launch by launch: the definition of the chain ----
static int chain_by_launches(const wqaa_chain_item* items, int count, int m, hipStream_t stream) {
  auto out_of = [&](int i) -> void* { return items[i].C; };
  for (int i = 0; i < count; ++i) {
    const wqaa_chain_item& it = items[i];
    const void* in = it.input_from >= 0 ? out_of(it.input_from) : it.A;
    const void* res = it.residual_from >= 0 ? out_of(it.residual_from) : it.residual;
    wqaa_epilogue epi;
    memset(&epi, 0, sizeof(epi));
    epi.struct_size = (int32_t)sizeof(epi);
    int st;
    if (it.kind == 1) {
      wqaa_group_item g, u;
      memset(&g, 0, sizeof(g));
      memset(&u, 0, sizeof(u));
      g.desc = u.desc = it.desc;
      g.A = u.A = in;
      g.B = it.B; g.Scale = it.Scale; g.Zeros = it.Zeros; g.Bias = it.Bias;
      u.B = it.B2; u.Scale = it.Scale2; u.Zeros = it.Zeros2; u.Bias = it.Bias2;
      if (it.norm_weight) {
        epi.flags = WQAA_EPI_RMSNORM_INPUT;
        epi.norm_weight = it.norm_weight;
        epi.norm_eps = it.norm_eps;
      }
      st = wqaa_matmul_gate_up(&g, &u, out_of(i), m, stream, it.norm_weight ? &epi : nullptr);
    } else if (it.norm_weight || res) {
      if (it.norm_weight) {
        epi.flags = WQAA_EPI_RMSNORM_INPUT;
        epi.norm_weight = it.norm_weight;
        epi.norm_eps = it.norm_eps;
      } else {
        epi.flags = WQAA_EPI_ADD_RESIDUAL;
        epi.residual = res;
      }
      st = wqaa_matmul_ex(it.desc, in, it.B, nullptr, it.Scale, it.Zeros, it.Bias, out_of(i), m, stream, &epi);
    } else {
      st = wqaa_matmul(it.desc, in, it.B, nullptr, it.Scale, it.Zeros, it.Bias, out_of(i), m, stream);
    }
    if (st != WQAA_OK) return st;
  }
  return WQAA_OK;
}

int chain_plan(const wqaa_chain_item* items, int count, int m, int* launches, wqaa_plan* plan) {
  int st = chain_validate(items, count, m);
  if (st != WQAA_OK) return st;
  if (plan) memset(plan, 0, sizeof(*plan));
  if (launches) *launches = count;
  return WQAA_OK;
}

int chain_launch(const wqaa_chain_item* items, int count, int m, hipStream_t stream) {
  int st = chain_validate(items, count, m);
  if (st != WQAA_OK) return st;
  if (m == 0) return WQAA_OK;
  return chain_by_launches(items, count, m, stream);
}

}  // namespace wqaa
