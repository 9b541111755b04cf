// sgym_experiments.hpp -- instrumentation of experiment builds (tools/ab_build.sh <name> -DSG_PHASE_TIMERS /
// -DSG_RSS_STATS): cycle counters per phase of a step, statistics of the RSS line-test queue.  None of it changes a result;
// in the product build every macro below is empty and every struct has no members.  (The timing ABLATIONS -- builds that
// leave a phase out and produce wrong results on purpose -- are not in the sources at all: tools/experiments/ablations.patch.)
#pragma once

namespace sg {

// Experiment builds (-DSG_PHASE_TIMERS, tools/ab_build.sh): where do the cycles of a step go?  PH(i) closes phase i: the
// cycles since the previous mark are added to counter i (wave-uniform scalar work); flushed once at the end of the kernel.
#ifdef SG_PHASE_TIMERS
struct PhaseTimers {
    unsigned long long acc[16], last;
    __device__ __forceinline__ void start() { for (int i = 0; i < 16; ++i) acc[i] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[i] += now - last; last = now; }
    __device__ __forceinline__ void flush(unsigned long long *out) { if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; ++i) if (acc[i]) atomicAdd(out + i, acc[i]); }
};
#define PH(i) ptm.mark(i)
#else
struct PhaseTimers {};
#define PH(i) ((void)0)
#endif

#ifdef SG_RSS_STATS
static __device__ unsigned long long sg_rss_stats[8]; // experiment builds: flushes, groups, items, passes, updates (per wavefront)
#define RSS_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&sg_rss_stats[i], (unsigned long long)(v)); } while (0)
#else
#define RSS_STAT(i, v) ((void)0)
#endif

} // namespace sg
