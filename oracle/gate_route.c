/* CPU restatement (TEST INFRASTRUCTURE, see oracle/__init__.py) of the integer /
 * index part of the M3ViT MoE hot path, with the arithmetic ORDER pinned so that
 * the HIP kernels can be checked bit-for-bit on expert indices and slots.
 *
 * gate:   NoisyGate_VMoE.forward, models/moe/ckpt/noisy_gate_vmoe.py:91-93,168,197-207
 * route:  the count/assign_pos step behind _fmoe_general_global_forward,
 *         models/moe/ckpt/custom_moe_layer.py:263-265 (same information as
 *         compute_gating, models/moe/moe.py:19-64), with a STABLE slot order.
 *
 * Pinned order (shared with m3vit_amd/csrc/gate.hip):
 *   logit[t][e] = fma chain over d = 0..D-1 starting from bias[e] (0 if none):
 *                 acc = fmaf(x[t][d], w[d*E+e], acc)
 *   noisy       = logit + noise[t][e] * std          (one mul, one add, no fma)
 *   selection   = k' = min(k+1, E) rounds of arg-max over the NOISY LOGITS,
 *                 ties -> lowest expert index.  The reference selects on the
 *                 softmax probabilities (:197-200); softmax is weakly monotone, so
 *                 this picks a valid top-k of the probabilities and is exact
 *                 whenever the probabilities have no ties (torch.topk's tie order
 *                 is unspecified).
 *   softmax     = m = max_e noisy; q_e = expf(noisy_e - m); s = q_0 + q_1 + ... (in
 *                 order); p_e = q_e / s.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define M3O_MAX_E 256

int m3o_gate_fwd(const float *x, int64_t T, int D, int64_t ldx,
                 const float *w, int E, const float *bias,
                 const float *noise, float std, int k,
                 int64_t *idx, float *score, float *top_logits,
                 float *clean, float *noisy_out, float *gates)
{
    if (E > M3O_MAX_E || E < 1 || k < 1 || k > E) return -1;
    int kp = (k + 1 < E) ? k + 1 : E;
    float lg[M3O_MAX_E], ns[M3O_MAX_E], p[M3O_MAX_E];
    unsigned char taken[M3O_MAX_E];
    for (int64_t t = 0; t < T; ++t) {
        const float *xr = x + t * ldx;
        for (int e = 0; e < E; ++e) {
            float acc = bias ? bias[e] : 0.0f;
            for (int d = 0; d < D; ++d) acc = fmaf(xr[d], w[(int64_t)d * E + e], acc);
            lg[e] = acc;
            float n = acc;
            if (noise && std != 0.0f) {
                float scaled = noise[t * E + e] * std;
                n = acc + scaled;
            }
            ns[e] = n;
        }
        float m = ns[0];
        for (int e = 1; e < E; ++e) m = ns[e] > m ? ns[e] : m;
        float s = 0.0f;
        for (int e = 0; e < E; ++e) { p[e] = expf(ns[e] - m); s = s + p[e]; }
        for (int e = 0; e < E; ++e) p[e] = p[e] / s;
        memset(taken, 0, (size_t)E);
        for (int j = 0; j < kp; ++j) {
            int best = -1;
            for (int e = 0; e < E; ++e) {
                if (taken[e]) continue;
                if (best < 0 || ns[e] > ns[best]) best = e;
            }
            taken[best] = 1;
            top_logits[t * kp + j] = p[best];
            if (j < k) { idx[t * k + j] = best; score[t * k + j] = p[best]; }
        }
        if (clean) memcpy(clean + t * E, lg, sizeof(float) * (size_t)E);
        if (noisy_out) memcpy(noisy_out + t * E, ns, sizeof(float) * (size_t)E);
        if (gates) {
            for (int e = 0; e < E; ++e) gates[t * E + e] = 0.0f;
            for (int j = 0; j < k; ++j) gates[t * E + idx[t * k + j]] = score[t * k + j];
        }
    }
    return 0;
}

/* idx[n] (n = T*k flat entries, values in [0,E)) -> counts[E], offsets[E+1],
 * pos[n] (slot of entry i in the expert-major buffer), row_of_slot[n] (inverse).
 * Slots inside an expert follow increasing flat entry index (stable). */
int m3o_route_build(const int64_t *idx, int64_t n, int E,
                    int64_t *counts, int64_t *offsets, int64_t *pos, int64_t *row_of_slot)
{
    for (int e = 0; e < E; ++e) counts[e] = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= E) return -1;
        counts[idx[i]]++;
    }
    offsets[0] = 0;
    for (int e = 0; e < E; ++e) offsets[e + 1] = offsets[e] + counts[e];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)E);
    if (!cur) return -2;
    for (int e = 0; e < E; ++e) cur[e] = offsets[e];
    for (int64_t i = 0; i < n; ++i) {
        int64_t s = cur[idx[i]]++;
        pos[i] = s;
        row_of_slot[s] = i;
    }
    free(cur);
    return 0;
}
