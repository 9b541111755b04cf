// sgym_grid.hpp -- BatchReplayEntity.add_entities stage 1: build_grid_kernel.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// ------------------------------------------------------------------------------------------------
// BatchReplayEntity.add_entities stage 1 (entity/batch.py:83-109): resample every batch-replay
// trajectory onto its scenario's union grid.  One thread per (grid row, entity slot).
// ------------------------------------------------------------------------------------------------
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ void build_grid_kernel(Params p, const int32_t *row_scen /*[totalN]*/, int64_t row0, int64_t row_end)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = row0 + gid / p.EP; // grid rows [row0, row_end): sg_upload launches one range per chunk of the knot copy
    int e = (int)(gid % p.EP);
    if (row >= row_end) return;
    int r = row_scen[row];
    uint32_t idx = (uint32_t)r * p.EP + e;
    const LanePtr st(p.stat + (size_t)(idx >> 6) * ST_COUNT * 64, (idx & 63) * 8u);
    int64_t meta = fld<int64_t>(st, ST_META);
    double out[6] = {0, 0, 0, 0, 0, 0};
    if (e < p.E && (meta & 0xff) == SG_KIND_REPLAY) {
        double tq = p.grid_t[row];
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        int n = (int)(meta >> 32);
        if (n == 1) { // batch.py:85-88: second knot at t + 0.1
            double x_lo = kn[0], x_hi = kn[0] + 1e-1;
            for (int c = 0; c < 6; ++c) {
                double v = kn[1 + c];
                if (tq < x_lo || tq > x_hi) out[c] = v;
                else {
                    // searchsorted_left over [x_lo, x_hi] clipped to 1 -> segment (0, 1)
                    double slope = (v - v) / (x_hi - x_lo);
                    out[c] = slope * (tq - x_lo) + v;
                }
            }
        } else if (tq < kn[0]) {
            for (int c = 0; c < 6; ++c) out[c] = kn[1 + c];
        } else if (tq > kn[(size_t)(n - 1) * 7]) {
            for (int c = 0; c < 6; ++c) out[c] = kn[(size_t)(n - 1) * 7 + 1 + c];
        } else {
            int lo = 0, hi = n;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (kn[(size_t)mid * 7] < tq) lo = mid + 1; else hi = mid;
            }
            int i1 = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
            const double *a = kn + (size_t)(i1 - 1) * 7, *b = kn + (size_t)i1 * 7;
            for (int c = 0; c < 6; ++c) {
                double slope = (b[1 + c] - a[1 + c]) / (b[0] - a[0]);
                out[c] = slope * (tq - a[0]) + a[1 + c];
            }
        }
    }
    for (int c = 0; c < 6; ++c) p.grid_y[((size_t)row * 6 + c) * p.EP + e] = out[c];
}
#endif // SG_UNIT_MAIN

} // namespace sg
