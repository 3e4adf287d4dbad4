// masks.hip -- the attention kernels' mask tables built ON THE DEVICE from the block-mask RULE (no L x L tensor, no host
// round trip): SURVEY.md section 8 f4.
//
// The reference regenerates generate_attention_mask(...) on the host every training step of the pretrain phase
// (models/dreamvla_model.py:610-628 -> 25-66: an (L, L) float tensor, uploaded as an nn.Parameter; the sdpa branch then
// expands it to (B,1,L,L)).  Round 1 did the same and then pulled the mask BACK to the host to derive the kernels' bit
// tables (two host round trips per step, nothing capturable).  The mask is a pure function of a handful of integers plus,
// for `mask_l_obs_ratio`, the obs-token columns drawn per window step with numpy's RNG (kept on the host so that the stream
// is consumed exactly as the reference consumes it) -- so the tables are computed from THOSE:
//
//   visible(r, c), r = (i, ro) query row of window step i, c = (j, co) key column of step j, blk = num_A + num_B:
//     act row  <=>  num_obs > 0 && aps > 0 && num_A + num_obs <= ro < num_A + num_obs + aps
//     atten_only_obs && act row :  j == i && ( 2 <= co < num_A  ||  obs column not dropped for step i  ||  (proprio && co == 1) )
//     otherwise                 :  (j <= i && co < num_A)  ||  (act row && j == i && co is an obs column)
//     atten_goal > 0 && atten_goal_state && ro is an obs row && i < K - atten_goal && c == (i + atten_goal) * blk + 1 : visible
//   (statement for statement what the loop at dreamvla_model.py:31-65 leaves behind; pinned bit for bit against
//   build_mask_tables(generate_attention_mask(...)) for the eight flag combinations of tests/test_mask.py).
//   Key compaction: a column is kept iff anybody can see it: co < num_A, or an obs column (when act rows exist) that is not
//   dropped for its step (the goal column has co == 1 < num_A).  Kept keys are listed leading columns first, obs columns
//   last (key_col below): the gather list may permute keys freely, and this order makes most 32 x 32 tiles uniform.
//
// One C-ABI call, three tiny launches: key_index, the two bit tables, the tile map.
#include "common.h"
#include "../../include/dvla.h"

namespace {

struct MaskRule {
  int K, num_A, num_B, num_obs, aps;
  int atten_goal, atten_goal_state, atten_only_obs, proprio;
  int n_drop;                 // dropped obs columns per window step (0 unless atten_only_obs && mask_l_obs_ratio > 0)
  const int32_t* drop;        // [K][n_drop] obs-token indices, device
  int L, Lk, per_step;        // L = K * blk; per_step = kept columns per window step; Lk = K * per_step
};

__device__ __forceinline__ bool has_act_rows(const MaskRule& m) { return m.num_obs > 0 && m.aps > 0; }
__device__ __forceinline__ bool dropped(const MaskRule& m, int step, int obs) {
  for (int t = 0; t < m.n_drop; ++t)
    if (m.drop[step * m.n_drop + t] == obs) return true;
  return false;
}
// compacted key k -> original column.  Order: the num_A leading columns of every window step first (step by step), then the
// undropped obs columns (step by step): columns with the same audience share 32-key tiles (see ops.build_mask_tables).
__device__ __forceinline__ int key_col(const MaskRule& m, int k) {
  const int blk = m.num_A + m.num_B;
  const int n_lead = m.K * m.num_A;
  if (k < n_lead) {
    const int j = k / m.num_A;
    return j * blk + (k - j * m.num_A);
  }
  const int per_obs = m.per_step - m.num_A;   // > 0 here
  const int ko = k - n_lead;
  const int j = ko / per_obs;
  int want = ko - j * per_obs;                // the want-th obs column of step j that is not dropped
  for (int o = 0; o < m.num_obs; ++o) {
    if (dropped(m, j, o)) continue;
    if (want == 0) return j * blk + m.num_A + o;
    --want;
  }
  return j * blk;             // unreachable for a consistent rule
}
__device__ __forceinline__ bool visible(const MaskRule& m, int r, int c) {
  const int blk = m.num_A + m.num_B;
  const int i = r / blk, ro = r - i * blk, j = c / blk, co = c - j * blk;
  const bool act_row = has_act_rows(m) && ro >= m.num_A + m.num_obs && ro < m.num_A + m.num_obs + m.aps;
  const bool obs_col = co >= m.num_A && co < m.num_A + m.num_obs;
  bool v;
  if (m.atten_only_obs && act_row) {
    v = j == i && ((co >= 2 && co < m.num_A) || (obs_col && !dropped(m, i, co - m.num_A)) || (m.proprio && co == 1));
  } else {
    v = (j <= i && co < m.num_A) || (act_row && j == i && obs_col);
  }
  if (m.num_obs > 0 && m.atten_goal > 0 && m.atten_goal_state && ro >= m.num_A && ro < m.num_A + m.num_obs &&
      i < m.K - m.atten_goal && c == (i + m.atten_goal) * blk + 1)
    v = true;
  return v;
}

__global__ void key_index_kernel(MaskRule m, int32_t* key_index) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m.Lk) key_index[k] = key_col(m, k);
}
// bits_q[r][kt]: bit b <=> compacted key 32 kt + b is visible to query r;  bits_k[k][qt]: bit b <=> query 32 qt + b sees key k
__global__ void bits_kernel(MaskRule m, const int32_t* __restrict__ key_index, uint32_t* bits_q, uint32_t* bits_k, int nkt, int nqt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nq_words = m.L * nkt;
  if (idx < nq_words) {
    const int r = idx / nkt, kt = idx - r * nkt;
    uint32_t w = 0;
    for (int b = 0; b < 32; ++b) {
      const int k = 32 * kt + b;
      if (k < m.Lk && visible(m, r, key_index[k])) w |= 1u << b;
    }
    bits_q[idx] = w;
  } else if (idx < nq_words + m.Lk * nqt) {
    const int t = idx - nq_words;
    const int k = t / nqt, qt = t - k * nqt;
    const int c = key_index[k];
    uint32_t w = 0;
    for (int b = 0; b < 32; ++b) {
      const int r = 32 * qt + b;
      if (r < m.L && visible(m, r, c)) w |= 1u << b;
    }
    bits_k[t] = w;
  }
}
// tile_map[qt][kt]: 0 = nothing visible (skipped), 1 = every valid (query, key) pair visible, 2 = mixed
__global__ void tile_map_kernel(MaskRule m, const uint32_t* __restrict__ bits_q, uint8_t* tile_map, int nkt, int nqt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nqt * nkt) return;
  const int qt = idx / nkt, kt = idx - qt * nkt;
  const int rows = (m.L - 32 * qt) < 32 ? (m.L - 32 * qt) : 32;
  const int keys = (m.Lk - 32 * kt) < 32 ? (m.Lk - 32 * kt) : 32;
  int vis = 0;
  for (int r = 0; r < rows; ++r) vis += __popc(bits_q[(32 * qt + r) * nkt + kt]);
  tile_map[idx] = vis == 0 ? 0 : (vis == rows * keys ? 1 : 2);
}

}  // namespace

extern "C" int dvla_mask_tables(const dvla_mask_rule* q, const int32_t* drop, int32_t* key_index, uint32_t* bits_q,
                                uint32_t* bits_k, uint8_t* tile_map, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!q || !key_index || !bits_q || !bits_k || !tile_map) return DVLA_ERR_ARG;
  if (q->K < 1 || q->num_A < 2 || q->num_B < 0 || q->num_obs < 0 || q->action_pred_steps < 0 || q->n_drop < 0) return DVLA_ERR_ARG;
  if (q->num_obs + q->action_pred_steps > q->num_B || q->n_drop > q->num_obs || (q->n_drop > 0 && !drop)) return DVLA_ERR_ARG;
  MaskRule m;
  m.K = q->K; m.num_A = q->num_A; m.num_B = q->num_B; m.num_obs = q->num_obs; m.aps = q->action_pred_steps;
  m.atten_goal = q->atten_goal; m.atten_goal_state = q->atten_goal_state; m.atten_only_obs = q->atten_only_obs;
  m.proprio = q->attn_robot_proprio_state;
  m.n_drop = (q->atten_only_obs && q->num_obs > 0 && q->action_pred_steps > 0) ? q->n_drop : 0;
  m.drop = drop;
  m.L = q->K * (q->num_A + q->num_B);
  const bool obs_keys = q->num_obs > 0 && q->action_pred_steps > 0;
  m.per_step = q->num_A + (obs_keys ? q->num_obs - m.n_drop : 0);
  m.Lk = q->K * m.per_step;
  const int nkt = (m.Lk + 31) / 32, nqt = (m.L + 31) / 32;
  hipLaunchKernelGGL(key_index_kernel, dim3((m.Lk + 255) / 256), dim3(256), 0, stream, m, key_index);
  const int words = m.L * nkt + m.Lk * nqt;
  hipLaunchKernelGGL(bits_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, m, key_index, bits_q, bits_k, nkt, nqt);
  hipLaunchKernelGGL(tile_map_kernel, dim3((nqt * nkt + 255) / 256), dim3(256), 0, stream, m, bits_q, tile_map, nkt, nqt);
  return dvla_check_launch();
}
