// dsp_cluster_protocol.h -- the two lock-free functions of the clustered LSTM launches (dsp_lstmc_kernel, dsp_kernels.hip):
// the ADMISSION of a cluster's member workgroups and the WAIT for the members' arrivals at a time step.  They are written
// against an `Ops` policy (the atomic operations on the cluster's two counters, a pause, a clock) so that the very same source
// is compiled
//   * into the gfx950 kernels (Ops = relaxed agent-scope atomics on device memory, s_sleep, s_memtime), and
//   * into tests/native/cluster_model.cpp (Ops = std::atomic, threads under ThreadSanitizer: members that stall, are never
//     scheduled, abandon mid-step),
// instead of the test holding a transcription that can drift from the kernel.
//
// The counters of one cluster (a site tile x direction of one layer launch), both zeroed by the forward's first launch:
//   arrivals  -- counts the hand-offs of h: a member (round 4: one lane per workgroup; round 5: one lane per wave) adds 1 after
//                its write-through h stores have been acknowledged; consumers wait for P x per-step arrivals x (step + 1);
//   state     -- the members counted in so far, and bit 31 = the cluster was ABANDONED (to the clean-up launch behind this
//                one, which recomputes it from scratch).  One word for both, so that "all P arrived" and "abandoned" exclude
//                each other: a member gives the cluster up with a CAS on the value it saw, which fails if anybody arrived.
#ifndef DSP_CLUSTER_PROTOCOL_H
#define DSP_CLUSTER_PROTOCOL_H

#if defined(__HIPCC__)   // (both passes of hipcc: the callers are __device__ functions)
#define DSP_CP_FN __device__ __forceinline__
#else
#define DSP_CP_FN inline
#endif

constexpr unsigned kClusterAbandon = 0x80000000u;

// Wait until `target` arrivals have been counted (every wave polls for itself: no workgroup barrier inside the k-loop).
// Returns false when the cluster was given up: a member that sees no progress for Ops::kSpinLimit polls (seconds on the GPU: a
// member that never became resident or died -- admission makes that all but impossible) marks the cluster abandoned, every
// member notices within kCheckMask + 1 polls and leaves, and the clean-up launch computes the cluster from scratch (it
// recomputes every step, so what the members had written is overwritten).  Until round 5 this was a trap: one stuck poll
// ended the process -- with 8 ranks, the job.
template <class Ops>
DSP_CP_FN bool dsp_wait_arrivals(Ops ops, unsigned target) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = ops.load_arrivals();
        if (v >= target) return true;
        ops.pause();
        if ((spins & Ops::kCheckMask) == Ops::kCheckMask) {
            const unsigned st = ops.load_state();
            if (st & kClusterAbandon) return false;
            if (spins > Ops::kSpinLimit) {
                ops.or_state(kClusterAbandon);
                return false;
            }
        }
    }
}

// Admission of a cluster (one lane of every member workgroup, before anything else): the members wait for each other at every
// step, so ALL of them must be resident -- which the dispatcher does not promise once other launches compete for the CUs
// (five concurrent clustered dispatches sharing an XCD's 32 slots evenly hold 6 members of 8 each: nobody ever completes).
// Each member counts itself in and waits until all P are there; a member that has waited `limit` ticks ABANDONS the cluster
// for all (one CAS on the word that also holds the count): every member, present or still to come, exits at once, and the
// clean-up launch behind this one (the workgroup-local form, no waiting between workgroups) computes the abandoned clusters.
// Returns true when the cluster runs.
template <class Ops>
DSP_CP_FN bool dsp_cluster_admit(Ops ops, unsigned P, unsigned long long limit) {
    unsigned s = ops.load_state();
    for (;;) {
        if (s & kClusterAbandon) return false;
        if (ops.cas_state(s, s + 1u)) break;   // (a failed CAS leaves the current value in s)
    }
    const unsigned long long t0 = ops.now();
    for (;;) {
        s = ops.load_state();
        if (s & kClusterAbandon) return false;
        if ((s & 0xffffu) >= P) return true;
        if (ops.now() - t0 > limit) {
            if (ops.cas_state(s, s | kClusterAbandon)) return false;
            continue;   // (the word moved: somebody arrived or abandoned meanwhile -- look again)
        }
        ops.pause();
    }
}

#endif
