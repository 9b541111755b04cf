// pair_bench_pairs.hpp -- crowd_pair for N pairs side by side (the interleaved form pair_bench.hip times against the plain one).
// Lived in sgym_crowd.hpp while the round-4/5 walker kernels used it; the product evaluates one pair per loop round
// (SG_CROWD_ILP = 1 measured fastest), so the N-way form is kept here, with the micro-benchmark that is its only user.
#pragma once
namespace sg {
// crowd_pair for N pairs at once, statement by statement ACROSS the pairs: the instruction stream alternates between N
// independent dependency chains.  A pair is one chain of ~130 dependent fp64 operations (an fp64 result can feed the next
// instruction only ~14 cycles after its issue, 4 cycles apart is the issue rate): written pair after pair the chains stay
// apart in the stream and a wavefront that is alone on its SIMD (sgym_walk.hpp) runs at the latency, not at the issue rate.
// Same operations in the same order per pair: the same bits as crowd_pair.
#define SG_EACH(u) _Pragma("unroll") for (int u = 0; u < N; ++u)
template <int N>
__device__ __forceinline__ void sg_sqrt_core_n(const double (&x)[N], double (&out)[N])
{
    double y[N], g[N], h[N], r[N], d[N];
    SG_EACH(u) y[u] = __builtin_amdgcn_rsq(x[u]);
    SG_EACH(u) { g[u] = x[u] * y[u]; h[u] = y[u] * 0.5; }
    SG_EACH(u) r[u] = __builtin_fma(-h[u], g[u], 0.5);
    SG_EACH(u) { g[u] = __builtin_fma(g[u], r[u], g[u]); h[u] = __builtin_fma(h[u], r[u], h[u]); }
    SG_EACH(u) d[u] = __builtin_fma(-g[u], g[u], x[u]);
    SG_EACH(u) g[u] = __builtin_fma(d[u], h[u], g[u]);
    SG_EACH(u) d[u] = __builtin_fma(-g[u], g[u], x[u]);
    SG_EACH(u) out[u] = __builtin_fma(d[u], h[u], g[u]);
}
// the refined reciprocal of RecipDiv (its b-only part), N at once
template <int N>
__device__ __forceinline__ void sg_recip_n(const double (&den)[N], double (&r)[N])
{
    double r0[N], e0[N], r1[N], e1[N];
    SG_EACH(u) r0[u] = __builtin_amdgcn_rcp(den[u]);
    SG_EACH(u) e0[u] = __builtin_fma(-den[u], r0[u], 1.0);
    SG_EACH(u) r1[u] = __builtin_fma(r0[u], e0[u], r0[u]);
    SG_EACH(u) e1[u] = __builtin_fma(-den[u], r1[u], 1.0);
    SG_EACH(u) r[u] = __builtin_fma(r1[u], e1[u], r1[u]);
}
// RecipDiv::div with the reciprocal r of b: q0 = a r, e = fma(-b, q0, a), q = fma(e, r, q0)
template <int N>
__device__ __forceinline__ void sg_rdiv_n(const double (&a)[N], const double (&b)[N], const double (&r)[N], double (&q)[N])
{
    double q0[N], e[N];
    SG_EACH(u) q0[u] = a[u] * r[u];
    SG_EACH(u) e[u] = __builtin_fma(-b[u], q0[u], a[u]);
    SG_EACH(u) q[u] = __builtin_fma(e[u], r[u], q0[u]);
}
template <int N>
__device__ __forceinline__ void crowd_pair_n(const CrowdConsts &C, const double (&rx)[N], const double (&ry)[N], const double (&odx)[N],
                                             const double (&ody)[N], const double (&sx)[N], const double (&sy)[N], const double (&ss)[N],
                                             double (&c1x)[N], double (&c1y)[N], double (&c2x)[N], double (&c2y)[N], double (&d2)[N],
                                             bool (&bad)[N])
{
    double rxx[N], a_rn[N], rn[N], qx[N], qy[N], a_qn[N], qn[N], sum[N], a_b[N], b[N], rb[N], k1[N], rrn[N], rqn[N];
    double rxn[N], ryn[N], qxn[N], qyn[N], dbx[N], dby[N], one[N], inv_b[N], xarg[N], ex[N], k2[N], repx[N], repy[N], a_rep[N], m[N], a[N];
    SG_EACH(u) rxx[u] = rx[u] * rx[u];
    SG_EACH(u) { d2[u] = rxx[u] + ry[u] * ry[u]; a_rn[u] = __builtin_fma(ry[u], ry[u], rxx[u]); }
    sg_sqrt_core_n<N>(a_rn, rn);
    SG_EACH(u) { qx[u] = rx[u] - sx[u]; qy[u] = ry[u] - sy[u]; }
    SG_EACH(u) a_qn[u] = __builtin_fma(qy[u], qy[u], qx[u] * qx[u]);
    sg_sqrt_core_n<N>(a_qn, qn);
    SG_EACH(u) qn[u] = qn[u] + 0.0000000001;
    SG_EACH(u) sum[u] = rn[u] + qn[u];
    SG_EACH(u) a_b[u] = sum[u] * sum[u] - ss[u];
    sg_sqrt_core_n<N>(a_b, b);
    SG_EACH(u) b[u] = (1.0 / 2) * b[u];
    sg_recip_n<N>(b, rb);
    SG_EACH(u) one[u] = 1.0;
    sg_rdiv_n<N>(one, b, rb, inv_b);
    SG_EACH(u) k1[u] = (1.0 / 4) * inv_b[u] * sum[u];
    sg_recip_n<N>(rn, rrn);
    sg_recip_n<N>(qn, rqn);
    sg_rdiv_n<N>(rx, rn, rrn, rxn);
    sg_rdiv_n<N>(ry, rn, rrn, ryn);
    sg_rdiv_n<N>(qx, qn, rqn, qxn);
    sg_rdiv_n<N>(qy, qn, rqn, qyn);
    SG_EACH(u) { dbx[u] = k1[u] * (rxn[u] + qxn[u]); dby[u] = k1[u] * (ryn[u] + qyn[u]); }
    // rsig.div(-b): the shared reciprocal of sigma
    {
        double q0[N], e[N];
        SG_EACH(u) q0[u] = -b[u] * C.sig_r;
        SG_EACH(u) e[u] = __builtin_fma(-C.sig_b, q0[u], -b[u]);
        SG_EACH(u) xarg[u] = __builtin_fma(e[u], C.sig_r, q0[u]);
    }
    // crowd_exp, N at once
    {
        const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10, INVLN2 = 1.44269504088896338700e+00;
        const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                     P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
        double k[N], hi[N], lo[N], r[N], t[N], c[N], den[N], rd[N], rc[N], q[N], y[N];
        SG_EACH(u) k[u] = __builtin_rint(xarg[u] * INVLN2);
        SG_EACH(u) { hi[u] = xarg[u] - k[u] * LN2HI; lo[u] = k[u] * LN2LO; }
        SG_EACH(u) r[u] = hi[u] - lo[u];
        SG_EACH(u) t[u] = r[u] * r[u];
        SG_EACH(u) c[u] = P4 + t[u] * P5;
        SG_EACH(u) c[u] = P3 + t[u] * c[u];
        SG_EACH(u) c[u] = P2 + t[u] * c[u];
        SG_EACH(u) c[u] = P1 + t[u] * c[u];
        SG_EACH(u) c[u] = r[u] - t[u] * c[u];
        SG_EACH(u) den[u] = 2.0 - c[u];
        sg_recip_n<N>(den, rd);
        SG_EACH(u) rc[u] = r[u] * c[u];
        sg_rdiv_n<N>(rc, den, rd, q);
        SG_EACH(u) y[u] = 1.0 - ((lo[u] - q[u]) - hi[u]);
        SG_EACH(u) { const double e_ = ldexp(y[u], (int)k[u]); ex[u] = xarg[u] < -745.13321910194110842 ? 0.0 : e_; }
    }
    SG_EACH(u) k2[u] = C.k2_scale * ex[u];
    SG_EACH(u) { repx[u] = k2[u] * dbx[u]; repy[u] = k2[u] * dby[u]; }
    SG_EACH(u) { c2x[u] = C.k3 * rx[u]; c2y[u] = C.k3 * ry[u]; }
    SG_EACH(u) a_rep[u] = __builtin_fma(repy[u], repy[u], repx[u] * repx[u]);
    sg_sqrt_core_n<N>(a_rep, m);
    SG_EACH(u) m[u] = m[u] + 0.0000000001;
    SG_EACH(u) a[u] = __builtin_fma(ody[u], repy[u], odx[u] * repx[u]);
    SG_EACH(u) {
        const double cm = C.cos_sight * m[u], slack = __builtin_fabs(cm) * 0x1p-50;
        const bool yes = a[u] >= cm + slack, no = a[u] <= cm - slack;
        const double w1 = yes ? 1.0 : C.sight_weight;
        c1x[u] = w1 * repx[u];
        c1y[u] = w1 * repy[u];
        const int h123 = min(min(__double2hiint(a_rn[u]), __double2hiint(a_qn[u])), __double2hiint(a_b[u]));
        bad[u] = !((h123 >= 0x39B00000) & (__double2hiint(a_rep[u]) >= 0x14300000) & (yes | no));
    }
}

#undef SG_EACH
} // namespace sg
