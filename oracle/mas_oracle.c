/*
 * mas_oracle.c — CPU ORACLE (test infrastructure; never linked or imported by the product path).
 *
 * Plain-C restatement of the reference's monotonic alignment search (MAS), used by tests/ as the checker
 * for the HIP kernel and by bench.py's cpu_baseline leg.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline may call this.
 *
 * Follows, statement by statement:
 *   glow_tts_train/monotonic_align/core.pyx:9-35   maximum_path_each  (forward DP in place + backtrack)
 *   glow_tts_train/monotonic_align/core.pyx:40-45  maximum_path_c     (serial loop over the batch; the
 *                                                   reference's prange has no -fopenmp, setup.py:9-13)
 * Arithmetic contract (core.c:2697-2703): one fp32 add per cell, max(a,b) == (v_prev > v_cur) ? v_prev : v_cur,
 * max_neg_val = -1e9f.  Built with -ffp-contract=off so no fused multiply-add can change a bit.
 *
 * Parity pinning: checked bit-for-bit against the reference's own Cython kernel (oracle/_ref, built from
 * /root/reference by oracle/Makefile) in tests/test_oracle_mas.py, and against tests/golden/mas_*.npz
 * generated from that kernel by oracle/make_golden.py.
 *
 * One defined deviation: the reference reads value[index, -1] (out of bounds) when t_x > t_y;
 * here the y == 0 column never looks left (the move-up test is skipped), which is what the reference does
 * for every input it is ever given (t_x <= t_y).
 */
#include <stdint.h>

#define MAS_MAX_NEG (-1e9f)

/* value: [t_x_stride rows][t_y_stride] fp32, mutated in place (cumulative scores inside the band)
 * path : same shape int32, must be zeroed by the caller (monotonic_align/__init__.py:15) */
static void mas_each(int32_t *path, float *value, int ld, int t_x, int t_y, float max_neg_val)
{
    int x, y;
    float v_prev, v_cur;
    int index = t_x - 1;

    for (y = 0; y < t_y; ++y) {
        int lo = t_x + y - t_y; if (lo < 0) lo = 0;
        int hi = y + 1;         if (hi > t_x) hi = t_x;
        for (x = lo; x < hi; ++x) {
            if (x == y) v_cur = max_neg_val;
            else        v_cur = value[(long)x * ld + (y - 1)];
            if (x == 0) {
                if (y == 0) v_prev = 0.0f;
                else        v_prev = max_neg_val;
            } else {
                v_prev = value[(long)(x - 1) * ld + (y - 1)];
            }
            {
                float m = (v_prev > v_cur) ? v_prev : v_cur;
                value[(long)x * ld + y] = m + value[(long)x * ld + y];
            }
        }
    }

    for (y = t_y - 1; y > -1; --y) {
        if (index < 0) break;               /* t_x == 0: nothing to mark */
        path[(long)index * ld + y] = 1;
        if (index != 0 && y > 0 &&
            (index == y || value[(long)index * ld + (y - 1)] < value[(long)(index - 1) * ld + (y - 1)]))
            index = index - 1;
    }
}

/* paths, values: [b][t_x][t_y] C-contiguous.  Mirrors maximum_path_c(paths, values, t_xs, t_ys, -1e9). */
void mas_oracle_batch(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys,
                      int b, int t_x_max, int t_y_max)
{
    int i;
    for (i = 0; i < b; ++i) {
        long off = (long)i * t_x_max * t_y_max;
        mas_each(paths + off, values + off, t_y_max, t_xs[i], t_ys[i], MAS_MAX_NEG);
    }
}
