// nbody_kernels_f64.hip — fp64 step kernel for the reference's own inputs (testcases/b*.in, n <= 1024, |q| ~ 3e20 m:
// fp32 cannot even represent r2), with the scenario monitors evaluated on the GPU.
//
// One launch = one run_step (samples/nbody.cc:51-89): accelerations from the old positions, kick, drift — and,
// in front of it, the O(1) monitor that main() evaluates after the PREVIOUS step (nbody.cc:118-121,131-137;
// hw5.cu:241-309), so a scenario is one launch per step with no extra <<<1,1>>> kernels and no host round trip.
//
// Replaces compute_accelerations_gpu (hw5.cu:159-215, thread per pair + 3 fp64 global atomics),
// update_positions_gpu (hw5.cu:231-239), clear_a_gpu (hw5.cu:224-229), calc_sq_min_dist_gpu (hw5.cu:241-252),
// calc_hit_time_step_gpu (hw5.cu:254-263), problem3_preprocess_gpu (hw5.cu:265-287), missile_cost_gpu
// (hw5.cu:289-309).
//
// Mapping: n is tiny, so the j-range of each target is split across S lanes of one wave (S = 1..64) and the S
// partial accelerations are combined with wave-level __shfl_xor reductions; S is chosen so that n*S threads
// fill the chip.  Sources are staged through LDS as SoA planes (x,y,z,G*m_eff), 1024 per pass (4 per thread, all
// loads of a pass in flight together and issued ahead of the monitor): lanes of a wave then read S consecutive
// doubles per plane (each broadcast to 64/S lanes) — conflict-free ds_read_b64.  The launch is latency-bound
// (n <= 1024 -> one pass, one barrier), so the structure minimises dependent memory round trips, not flops.
// Owner-computes, no atomics: results are bitwise reproducible run to run (the reference's are not).
//
// Positions ping-pong (qin -> qout) because other workgroups still read the old positions while this one
// updates its targets; velocities are owned by one lane each and updated in place.
#include "nbody_kernels.h"

namespace nbk {

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x) to fp64 rounding from the hardware seed: v_rsq_f64 (relative error e0 ~ 2^-26) followed by ONE third-order
// (Halley) correction y <- y*(1 + e/2 + 3e^2/8), e = 1 - x*y*y, whose remaining error ~ (5/16)*e0^3 is far below 2^-53.
// 6 fp64 VALU ops + the 16-cycle transcendental instead of the ~25-instruction IEEE sqrt + divide expansion the
// compiler emits for 1.0/sqrt(x) — the pair loop is ~40 % shorter.  (the testcase outputs do not
// move a digit for relative force errors up to 1e-7; this is at 1e-16.)
__device__ __forceinline__ double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    double t = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, t, y);
}

// lane l receives the value of lane l+K of its 16-lane row (0 past the row end): v_mov_b32 with a DPP row_shl modifier —
// a plain VALU move, no LDS crossbar round trip like ds_bpermute (__shfl_*).
template <int K>
__device__ __forceinline__ double row_shl_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + K, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + K, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double dist2_bodies(const double* q, int n, int i, int j) {
    double dx = q[i] - q[j];
    double dy = q[n + i] - q[n + j];
    double dz = q[2 * n + i] - q[2 * n + j];
    return dx * dx + dy * dy + dz * dz;  // same association as nbody.cc:134 / hw5.cu:258
}

// The O(1) monitors of a step (the reference's <<<1,1>>> kernels, hw5.cu:241-309) on the state with index step-1: P1 min
// d^2, P2 first hit, missile arrival per watched device.  Pure function of HBM state and kernel arguments, so every caller
// that evaluates it reaches the same decisions; `rec` says whether this lane records them.  A value another workgroup of
// THIS launch may already have written (arrival_step == step-1, hit_step == step-1) leads to the same decision as
// re-deriving it.
__device__ __forceinline__ void monitor_f64(const F64Args& a, const F64Scenario& sc, int n, int step, bool rec, int& skip,
                                            unsigned& destroyed, unsigned& snap) {
    if (sc.kind >= 0) {
        const int idx = step - 1;
        F64Monitor* mon = a.mon;
        // every load the monitor can need is issued up front, unconditionally, so that thread 0 pays ONE memory round
        // trip — the whole workgroup waits for its decisions at the barrier below (P2 and the Problem-3 runs are the
        // critical chain of the program).  Devices beyond the first PRE watched ones take the dependent path.
        constexpr int PRE = 4;
        const double d2 = dist2_bodies(a.qin, n, sc.planet, sc.asteroid);
        int hit = -2, arr_pre[PRE];
        double m_pre[PRE], d2_pre[PRE];
        if (sc.kind != 0) {
            hit = mon->hit_step;
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const bool on = k < sc.n_watch;
                const int d = on ? sc.watch[k] : sc.planet;
                arr_pre[k] = on ? mon->arrival_step[k] : -2;
                m_pre[k] = a.m[d];
                d2_pre[k] = dist2_bodies(a.qin, n, sc.planet, d);
            }
        }
        if (sc.kind == 0) {  // MIN_DIST: nbody.cc:118-121 (min of squares; sqrt on the host)
            if (rec && d2 < mon->min_d2) mon->min_d2 = d2;
        } else {
            if (hit == -2 && d2 < sc.R2) {  // nbody.cc:134-137 ; hw5.cu:295-298 (hit test comes first)
                hit = idx;
                if (rec) mon->hit_step = idx;
            }
            if (hit != -2) {
                skip = 1;  // P2 stops at the first hit; a destroyed-device run has failed
            } else {
                const double md = sc.missile_dstep * idx;  // hw5.cu:274,303
                for (int k = 0; k < sc.n_watch; ++k) {
                    const int d = sc.watch[k];
                    int arr;
                    double mk, dk2;
                    if (k < PRE) {
                        arr = arr_pre[k]; mk = m_pre[k]; dk2 = d2_pre[k];
                    } else {
                        arr = mon->arrival_step[k];
                        mk = a.m[d];
                        dk2 = (arr == -2 && mk != 0.0) ? dist2_bodies(a.qin, n, sc.planet, d) : 0.0;
                    }
                    if (arr == -2 && mk != 0.0 && dk2 < md * md) {  // hw5.cu:299 m[d] != 0
                        arr = idx;
                        if (rec) mon->arrival_step[k] = idx;
                    }
                    if (arr == idx) snap |= 1u << k;                                // hw5.cu:277-285
                    if (arr != -2 && sc.destroy_on_arrival) destroyed |= 1u << k;   // hw5.cu:306
                }
            }
        }
    }
}

#ifndef NB_K2_WG
#define NB_K2_WG 256  // threads per workgroup of the per-step kernel = 4 targets (one wave each) sharing one LDS stage of the
                      // system.  Measured optimum (profiles/r03_k2_wg_sweep.txt, n = 1024): 128 -> 6.56, 256 -> 5.26,
                      // 512 -> 5.44, 1024 -> 7.15 us/step
#endif
constexpr int K2_WG = NB_K2_WG;
constexpr int K2_TILE = 1024;  // sources staged per pass: 4 per thread, 32 KB of LDS; n <= 1024 needs ONE pass

// SELFCHECK: eps == 0 — the self pair must be skipped explicitly (r2 = 0); with eps > 0 it contributes s * 0 = +0 by
// itself, exactly like skipping it (nbody.cc:59), and the two compares + two selects per pair are saved
template <int S, bool SELFCHECK>
__device__ __forceinline__ void step_f64_body(const F64Args& a) {
    __shared__ double sx[K2_TILE], sy[K2_TILE], sz[K2_TILE], sg[K2_TILE];
    __shared__ int sh_skip;
    __shared__ unsigned sh_destroyed;  // bit k: watched device k has mass 0 for this step
    __shared__ unsigned sh_snap;       // bit k: snapshot state step-1 for watched device k now

    const int t = threadIdx.x;
    const int n = a.n;
    const F64Scenario& sc = a.scn;
    constexpr int PER = K2_TILE / K2_WG;                    // sources each thread stages per full pass
    const int per_n = n >= K2_TILE ? PER : (n + K2_WG - 1) / K2_WG;  // ... and for a small system (workgroup-uniform)

    // ---- first pass's source loads are issued BEFORE the monitor so that their latency and thread 0's dependent
    //      monitor chain (flag -> positions -> compare) overlap instead of adding up
    double lx[PER], ly[PER], lz[PER], lm[PER], lc[PER];
    auto issue_loads = [&](int base) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (u < per_n) {
                const int j = base + u * K2_WG + t;
                const int jc = j < n ? j : n - 1;
                lx[u] = a.qin[jc]; ly[u] = a.qin[n + jc]; lz[u] = a.qin[2 * n + jc];
                lm[u] = a.m[jc]; lc[u] = a.coef[jc];
            }
        }
    };
    issue_loads(0);
#if NB_STEP_STAMPS
    const bool stamping = a.stamps && blockIdx.x == 0 && t == 0;
    const unsigned long long t_entry = stamping ? wall_clock64() : 0;
    unsigned long long* stamp = nullptr;
#endif

    constexpr int TPB = K2_WG / S;  // targets per workgroup
    const int ls = t % S;        // this lane's slice of the source range
    const int i = blockIdx.x * TPB + t / S;
    const bool owner = (ls == 0) && (i < n);
    const int ic = i < n ? i : n - 1;
    // the target's own position and (owner lanes) velocity are requested now too: nothing below may add another
    // dependent memory round trip to the launch's critical path
    const double xi = a.qin[ic], yi = a.qin[n + ic], zi = a.qin[2 * n + ic];
    double vx0 = 0, vy0 = 0, vz0 = 0;
    if (owner) { vx0 = a.v[i]; vy0 = a.v[n + i]; vz0 = a.v[2 * n + i]; }

    // which step is this?  Eager launches carry it; a launch replayed from a graph derives it from the scenario's
    // device-resident control word (wave-uniform scalar loads, issued behind the vector loads above)
    int step = a.step, do_update = a.do_update;
    double fst = a.fst;
    if (a.ctl) {
        fst = a.fst_chunk[a.t];  // address known at capture time: in flight together with the control word, not behind it
        const F64Ctl ctl = *a.ctl;
        step = ctl.base_step + a.t;
        if (!ctl.active || step > a.last_step + 1) return;  // dormant slot / past the end: workgroup-uniform
        do_update = step <= a.last_step;
    }
#if NB_STEP_STAMPS
    if (stamping) {  // only launches that do work leave a record (idle nodes past the end of a run returned above)
        stamp = a.stamps + 2 * (size_t)((a.ctl ? a.t - 1 : a.step) % a.stamp_slots);
        stamp[0] = t_entry;
        stamp[1] = 0;
    }
#endif

    // ---- monitor on the state after step-1 (index step-1), evaluated identically by every workgroup;
    //      only workgroup 0 records it.  A value another workgroup of THIS launch may already have written
    //      (arrival_step == step-1, hit_step == step-1) leads to the same decision as re-deriving it.
    if (t == 0) {
        int skip = 0;
        unsigned destroyed = 0, snap = 0;
        monitor_f64(a, sc, n, step, blockIdx.x == 0, skip, destroyed, snap);
        if (!do_update) skip = 1;
        sh_skip = skip;
        sh_destroyed = destroyed;
        sh_snap = snap;
    }

    double ax = 0, ay = 0, az = 0;

    // stage a pass into LDS: G*m_eff with the device-mass law m0 + 0.5*m0*|sin(t/6000)| (nbody.cc:14-16,61-64), rounded
    // exactly like the CPU reference (no FMA contraction): coef = 0.5 for devices, 0 otherwise -> m + 0 = m.
    auto stage = [&](int base) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (u < per_n) {
                const int jl = u * K2_WG + t;
                const double mj = __dadd_rn(lm[u], __dmul_rn(__dmul_rn(lc[u], lm[u]), fst));
                sx[jl] = lx[u]; sy[jl] = ly[u]; sz[jl] = lz[u];
                sg[jl] = (base + jl < n) ? __dmul_rn(a.G, mj) : 0.0;  // G*mj, the reference's first product (nbody.cc:70)
            }
        }
    };
    stage(0);
    __syncthreads();  // monitor decisions + first pass visible
    const unsigned snap = sh_snap, destroyed = sh_destroyed;
    const bool skip = sh_skip != 0;

    // snapshot of (q,v) at missile arrival, taken from the not-yet-updated state (hw5.cu:277-285)
    if (snap && owner && a.snap_q) {
        for (int k = 0; k < sc.n_watch; ++k)
            if (snap & (1u << k)) {
                double* dq = a.snap_q + (size_t)k * 3 * n;
                double* dv = a.snap_v + (size_t)k * 3 * n;
                dq[i] = xi; dq[n + i] = yi; dq[2 * n + i] = zi;
                dv[i] = vx0; dv[n + i] = vy0; dv[2 * n + i] = vz0;
            }
    }
    if (skip) return;  // workgroup-uniform

    // a destroyed device (MISSILE scenario, one watched device) has mass 0 from the step after its arrival (hw5.cu:306):
    // its staged G*m is cleared (the staging thread and this one are ordered by the barriers around the write)
    int dead_j = -1;
    for (int k = 0; k < sc.n_watch; ++k)
        if (destroyed & (1u << k)) dead_j = sc.watch[k];

    for (int base = 0; base < n; base += K2_TILE) {
        if (base) {  // further passes (n > 1024)
            issue_loads(base);
            __syncthreads();  // previous pass fully consumed
            stage(base);
            __syncthreads();
        }
        if (dead_j >= base && dead_j < base + K2_TILE) {  // workgroup-uniform
            if (t == 0) sg[dead_j - base] = 0.0;
            __syncthreads();
        }
        const int lim = min(K2_TILE, n - base);
#pragma unroll 2
        for (int jj = ls; jj < lim; jj += S) {
            double dx = sx[jj] - xi;
            double dy = sy[jj] - yi;
            double dz = sz[jj] - zi;
            double r2 = dx * dx + dy * dy + dz * dz + a.eps2;
            double rinv = rsqrt_fast(r2);
            double s = sg[jj] * rinv * rinv * rinv;  // G*mj/(r2+eps2)^1.5
            if (SELFCHECK) s = (base + jj == i) ? 0.0 : s;  // j == i skipped (nbody.cc:59): needed for eps == 0 only
            ax += s * dx;
            ay += s * dy;
            az += s * dz;
        }
    }

    // wave-level reduction of the S partial accelerations of a target (adjacent lanes of one wave) into lane ls == 0:
    // across 16-lane rows with __shfl_xor (ds_bpermute), inside a row with DPP row_shl moves (plain VALU)
    if (S >= 64) { ax += shfl_xor_f64(ax, 32); ay += shfl_xor_f64(ay, 32); az += shfl_xor_f64(az, 32); }
    if (S >= 32) { ax += shfl_xor_f64(ax, 16); ay += shfl_xor_f64(ay, 16); az += shfl_xor_f64(az, 16); }
    if (S >= 16) { ax += row_shl_f64<8>(ax); ay += row_shl_f64<8>(ay); az += row_shl_f64<8>(az); }
    if (S >= 8) { ax += row_shl_f64<4>(ax); ay += row_shl_f64<4>(ay); az += row_shl_f64<4>(az); }
    if (S >= 4) { ax += row_shl_f64<2>(ax); ay += row_shl_f64<2>(ay); az += row_shl_f64<2>(az); }
    if (S >= 2) { ax += row_shl_f64<1>(ax); ay += row_shl_f64<1>(ay); az += row_shl_f64<1>(az); }

    if (owner) {
        if (a.acc_out) {
            a.acc_out[i] = ax; a.acc_out[n + i] = ay; a.acc_out[2 * n + i] = az;
        } else {
            // kick, then drift with the NEW velocity (nbody.cc:76-88), written without contraction
            double vx = __dadd_rn(vx0, __dmul_rn(ax, a.dt));
            double vy = __dadd_rn(vy0, __dmul_rn(ay, a.dt));
            double vz = __dadd_rn(vz0, __dmul_rn(az, a.dt));
            a.v[i] = vx; a.v[n + i] = vy; a.v[2 * n + i] = vz;
            a.qout[i] = __dadd_rn(xi, __dmul_rn(vx, a.dt));
            a.qout[n + i] = __dadd_rn(yi, __dmul_rn(vy, a.dt));
            a.qout[2 * n + i] = __dadd_rn(zi, __dmul_rn(vz, a.dt));
        }
    }
#if NB_STEP_STAMPS
    if (stamp) stamp[1] = wall_clock64();
#endif
}

template <int S, bool SELFCHECK>
__global__ __launch_bounds__(K2_WG) void nbody_step_f64(F64Args a) {
    step_f64_body<S, SELFCHECK>(a);
}

// One launch advances up to MAX_BATCH independent systems of the same n by one step each (blockIdx.y = system): the
// scenarios hw5 runs side by side (P3 per device: hw5.cu:587-588) then share ONE launch per step instead of contending
// for the command processor with one launch stream each.  Every system carries its own step index, |sin| and monitor.
template <int S, bool SELFCHECK>
__global__ __launch_bounds__(K2_WG) void nbody_step_f64_batched(F64BatchArgs b) {
    const F64Args& a = b.item[blockIdx.y];
    if (a.n <= 0) return;  // finished / not started: nothing to do for this slot
    step_f64_body<S, SELFCHECK>(a);
}

// ---------------------------------------------------------------------------------------------------------------
// K3 nbody_scenario_small_f64<S>: a whole scenario of a small system (n <= 128: testcases b20..b100) in ONE launch of
// ONE workgroup.  The reference — and K2 above — pay a kernel launch (plus monitor launches) per step: 200 000 steps x
// ~4 us is pure launch latency for systems whose 400..10 000 pairs take well under a microsecond to evaluate.
// Here the state lives in LDS (positions and G*m_eff ping-pong, 8 KB), velocities in the owner lanes' registers, the
// step loop runs inside the kernel with ONE s_barrier per step, and every thread evaluates the O(1) monitors redundantly
// from LDS (identical inputs -> identical, hence workgroup-uniform, decisions): no flags, no inter-workgroup protocol,
// no grid barrier, every wave reaches the loop exit.  Same arithmetic as K2 (same G*m_eff rounding, same pair term,
// same non-contracted kick/drift); only the summation split S differs.
// FEW: at most SMALL_FEW watched devices (the reference's inputs have 2-4) — their arrival steps live in registers and the
// monitor is straight-line code over them, with every LDS read it can need issued up front.  (A runtime-indexed arr[k] would
// live in scratch memory: measured +0.45 us per step at n = 20, +0.75 us at n = 100 for the FIRST_HIT / MISSILE scenarios.)
// !FEW: up to MAX_WATCH devices, same code unrolled over all of them.
constexpr int SMALL_FEW = 4;
template <int S, bool SELFCHECK, bool FEW>
__device__ __forceinline__ void scenario_small_body(const F64SmallArgs& a) {
    constexpr int NW = FEW ? SMALL_FEW : MAX_WATCH;
    __shared__ double sq[2][3][SMALL_N_MAX];
    __shared__ double sg[2][SMALL_N_MAX];
    const int t = threadIdx.x;
    const int n = a.n;
    const F64Scenario& sc = a.scn;
    const int ls = t % S;
    const int i = t / S;
    const bool owner = (ls == 0) && (i < n);
    const int ic = i < n ? i : n - 1;

    // owner state: velocity, mass law inputs
    double vx = 0, vy = 0, vz = 0, mi = 0, cmi = 0;
    if (owner) {
        vx = a.v[i]; vy = a.v[n + i]; vz = a.v[2 * n + i];
        mi = a.m[i];
        cmi = __dmul_rn(a.coef[i], mi);  // (0.5*m0) for devices, 0 otherwise
        sq[0][0][i] = a.q[i]; sq[0][1][i] = a.q[n + i]; sq[0][2][i] = a.q[2 * n + i];
        // G*m_eff for the first step to be taken
        sg[0][i] = __dmul_rn(a.G, __dadd_rn(mi, __dmul_rn(cmi, a.fst[a.first_step + 1])));
    }
    // scenario state, replicated in every thread (all derive it from the same LDS words)
    double min_d2 = a.mon->min_d2;
    int hit = a.mon->hit_step;
    int arr[NW];
    int dead_j = -1;  // MISSILE: the destroyed device's index once its missile has arrived
    unsigned alive = 0;  // bit k: watched device k has a non-zero base mass (hw5.cu:299), read once
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        arr[k] = a.mon->arrival_step[k];
        if (k < sc.n_watch && a.m[sc.watch[k]] != 0.0) alive |= 1u << k;
    }
    if (sc.destroy_on_arrival && sc.n_watch > 0 && arr[0] != -2) dead_j = sc.watch[0];
    if (owner && i == dead_j) sg[0][i] = 0.0;  // a destroyed device pulls nothing: its staged G*m is cleared (hw5.cu:306)
    __syncthreads();

    auto d2_of = [&](int buf, int p, int r) {
        double dx = sq[buf][0][p] - sq[buf][0][r];
        double dy = sq[buf][1][p] - sq[buf][1][r];
        double dz = sq[buf][2][p] - sq[buf][2][r];
        return dx * dx + dy * dy + dz * dz;
    };
    // monitor on the state with index idx held in buffer buf; returns true when the scenario must stop
    auto monitor = [&](int buf, int idx) -> bool {
        if (sc.kind < 0) return false;
        const double d2 = d2_of(buf, sc.planet, sc.asteroid);
        if (sc.kind == 0) {  // MIN_DIST  nbody.cc:118-121
            if (d2 < min_d2) min_d2 = d2;
            return false;
        }
        // planet-device distances of the devices still waiting for their missile: all LDS reads in flight together
        double dk[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k)
            dk[k] = (k < sc.n_watch && arr[k] == -2) ? d2_of(buf, sc.planet, sc.watch[k]) : 0.0;
        if (d2 < sc.R2) {  // nbody.cc:134-137 ; hw5.cu:295-298
            hit = idx;
            return true;
        }
        const double md = sc.missile_dstep * idx;
        const double md2 = md * md;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if (k < sc.n_watch && arr[k] == -2 && (alive & (1u << k)) && dk[k] < md2) {  // hw5.cu:299
                const int d = sc.watch[k];
                arr[k] = idx;
                if (sc.destroy_on_arrival) dead_j = d;  // hw5.cu:306
                if (a.snap_q && owner) {                // hw5.cu:277-285
                    double* dq = a.snap_q + (size_t)k * 3 * n;
                    double* dv = a.snap_v + (size_t)k * 3 * n;
                    dq[i] = sq[buf][0][i]; dq[n + i] = sq[buf][1][i]; dq[2 * n + i] = sq[buf][2][i];
                    dv[i] = vx; dv[n + i] = vy; dv[2 * n + i] = vz;
                }
            }
        }
        return false;
    };

    int cur = 0;
    int step = a.first_step + 1;
    bool stopped = false;
    // |sin| of the step after next is requested two iterations before it is needed: a global (L2) load takes about
    // as long as one whole iteration of this loop, so a one-deep prefetch would still stall the owners' update
    double fst_next = owner ? a.fst[step + 1] : 0.0;
    for (; step <= a.last_step; ++step) {
        const int dead_before = dead_j;
        if (monitor(cur, step - 1)) { stopped = true; break; }
        if (dead_j != dead_before) {  // the missile arrived at this state (workgroup-uniform, once per scenario)
            if (owner && i == dead_j) sg[cur][i] = 0.0;
            __syncthreads();
        }
        const double fst_after = owner ? a.fst[step + 2] : 0.0;
        const double xi = sq[cur][0][ic], yi = sq[cur][1][ic], zi = sq[cur][2][ic];
        double ax = 0, ay = 0, az = 0;
#pragma unroll 2
        for (int jj = ls; jj < n; jj += S) {
            double dx = sq[cur][0][jj] - xi;
            double dy = sq[cur][1][jj] - yi;
            double dz = sq[cur][2][jj] - zi;
            double r2 = dx * dx + dy * dy + dz * dz + a.eps2;
            double rinv = rsqrt_fast(r2);
            double s = sg[cur][jj] * rinv * rinv * rinv;
            if (SELFCHECK) s = (jj == i) ? 0.0 : s;  // j == i skipped (nbody.cc:59): with eps > 0 the pair is s * 0 = +0
            ax += s * dx;
            ay += s * dy;
            az += s * dz;
        }
        // the S = 8 lanes of a target are adjacent lanes of one 16-lane row and only lane ls == 0 needs the total:
        // three DPP row_shl folds (4, 2, 1)
        static_assert(S == 8, "reduction below is written for 8 lanes per target");
        ax += row_shl_f64<4>(ax); ay += row_shl_f64<4>(ay); az += row_shl_f64<4>(az);
        ax += row_shl_f64<2>(ax); ay += row_shl_f64<2>(ay); az += row_shl_f64<2>(az);
        ax += row_shl_f64<1>(ax); ay += row_shl_f64<1>(ay); az += row_shl_f64<1>(az);
        if (owner) {  // kick, drift (nbody.cc:76-88), then G*m_eff of the NEXT step into the other buffer
            vx = __dadd_rn(vx, __dmul_rn(ax, a.dt));
            vy = __dadd_rn(vy, __dmul_rn(ay, a.dt));
            vz = __dadd_rn(vz, __dmul_rn(az, a.dt));
            sq[cur ^ 1][0][i] = __dadd_rn(xi, __dmul_rn(vx, a.dt));
            sq[cur ^ 1][1][i] = __dadd_rn(yi, __dmul_rn(vy, a.dt));
            sq[cur ^ 1][2][i] = __dadd_rn(zi, __dmul_rn(vz, a.dt));
            sg[cur ^ 1][i] = (i == dead_j) ? 0.0 : __dmul_rn(a.G, __dadd_rn(mi, __dmul_rn(cmi, fst_next)));
        }
        __syncthreads();  // the only barrier of the step: buffer cur^1 complete, buffer cur free for step+1's writes
        cur ^= 1;
        fst_next = fst_after;
    }
    int done = stopped ? step - 1 : a.last_step;
    if (!stopped && a.final_monitor) (void)monitor(cur, a.last_step);

    if (owner) {
        a.q[i] = sq[cur][0][i]; a.q[n + i] = sq[cur][1][i]; a.q[2 * n + i] = sq[cur][2][i];
        a.v[i] = vx; a.v[n + i] = vy; a.v[2 * n + i] = vz;
    }
    if (t == 0) {
        a.mon->min_d2 = min_d2;
        a.mon->hit_step = hit;
#pragma unroll
        for (int k = 0; k < NW; ++k) a.mon->arrival_step[k] = arr[k];  // (slots beyond NW are not watched in this mode)
        *a.steps_done = done;
    }
}

template <int S, bool SELFCHECK, bool FEW>
__global__ __launch_bounds__(SMALL_N_MAX * S) void nbody_scenario_small_f64(F64SmallArgs a) {
    scenario_small_body<S, SELFCHECK, FEW>(a);
}

// Up to MAX_BATCH scenarios of equally sized small systems in ONE launch: workgroup k runs scenario k from its own
// state, with its own monitor, completely independently of the others (own LDS, no inter-workgroup traffic) — the whole
// reference program (P1, P2 and one Problem-3 run per device, hw5.cu:564-567,587-588) is then a single kernel launch on
// 2 + D compute units, with no host thread, stream or hardware queue per scenario.
template <int S, bool SELFCHECK, bool FEW>
__global__ __launch_bounds__(SMALL_N_MAX * S) void nbody_scenario_small_f64_batched(F64SmallBatchArgs b) {
    const F64SmallArgs& a = b.item[blockIdx.x];
    if (a.n <= 0) return;  // finished slot (workgroup-uniform)
    scenario_small_body<S, SELFCHECK, FEW>(a);
}

template <int S>
static int launch_small_s(const F64SmallArgs& a, int threads, hipStream_t stream) {
    const bool few = a.scn.n_watch <= SMALL_FEW;
    if (a.eps2 >= F64_EPS2_MIN) {
        if (few) hipLaunchKernelGGL((nbody_scenario_small_f64<S, false, true>), dim3(1), dim3(threads), 0, stream, a);
        else hipLaunchKernelGGL((nbody_scenario_small_f64<S, false, false>), dim3(1), dim3(threads), 0, stream, a);
    } else {
        if (few) hipLaunchKernelGGL((nbody_scenario_small_f64<S, true, true>), dim3(1), dim3(threads), 0, stream, a);
        else hipLaunchKernelGGL((nbody_scenario_small_f64<S, true, false>), dim3(1), dim3(threads), 0, stream, a);
    }
    return (int)hipGetLastError();
}

// Geometry: S = 8 lanes per target (three shuffle rounds), NPAD = n rounded up to a multiple of 8 targets per wave,
// threads = NPAD*S rounded up to whole waves: 256 threads for n <= 32 ... 1024 for n <= 128 — few waves keep the
// per-step barrier and the dependent chain (LDS read -> pair -> 3 shuffles -> update -> LDS write) short.
int launch_f64_small(const F64SmallArgs& a, hipStream_t stream) {
    if (a.n <= 0 || a.n > SMALL_N_MAX) return (int)hipErrorInvalidValue;
    constexpr int S = 8;
    int threads = ((a.n * S + 63) / 64) * 64;
    return launch_small_s<S>(a, threads, stream);
}

int launch_f64_small_batched(const F64SmallBatchArgs& b, int n, hipStream_t stream) {
    if (n <= 0 || n > SMALL_N_MAX || b.count <= 0 || b.count > MAX_BATCH) return (int)hipErrorInvalidValue;
    constexpr int S = 8;
    const int threads = ((n * S + 63) / 64) * 64;
    bool eps_positive = true, few = true;
    for (int k = 0; k < b.count; ++k) {
        eps_positive &= (b.item[k].n <= 0 || b.item[k].eps2 >= F64_EPS2_MIN);
        few &= (b.item[k].n <= 0 || b.item[k].scn.n_watch <= SMALL_FEW);
    }
    const dim3 grid(b.count), block(threads);
    if (eps_positive) {
        if (few) hipLaunchKernelGGL((nbody_scenario_small_f64_batched<S, false, true>), grid, block, 0, stream, b);
        else hipLaunchKernelGGL((nbody_scenario_small_f64_batched<S, false, false>), grid, block, 0, stream, b);
    } else {
        if (few) hipLaunchKernelGGL((nbody_scenario_small_f64_batched<S, true, true>), grid, block, 0, stream, b);
        else hipLaunchKernelGGL((nbody_scenario_small_f64_batched<S, true, false>), grid, block, 0, stream, b);
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// K1-f64 nbody_force_f64_large: the fp64 member of the large-N family (SURVEY §2.1 K1 "fp32, fp32+fp64-accumulate, fp64
// variants").  K2 stages the whole system in every workgroup and shares a target between lanes — right for n <= 1024,
// wasteful beyond.  Here, as in the fp32 K1: a lane owns R = 2 whole targets, the sources of a batch are wave-uniform
// and arrive through scalar loads (s_load_dwordx8 from each SoA plane) as SGPR operands of the fp64 VALU ops, the
// source range is sliced over blockIdx.y for enough workgroups, partial sums are combined by a reducer that also does
// the non-contracted kick-drift.  Pair cost: 3 add + 3 fma + v_rsq_f64 + 5 (Halley) + 3 mul + 3 fma = 17 fp64 VALU
// (4 cycles each) + 16 => ceiling 1024 SIMDs x 2.4 GHz x 64 / 84 = 1.87e12 pairs/s.
__global__ __launch_bounds__(WG) void nbody_gm_f64(const double* __restrict__ m, const double* __restrict__ coef,
                                                   double* __restrict__ gm, int n, double fst, double G) {
    const int j = blockIdx.x * WG + threadIdx.x;
    if (j < n) gm[j] = __dmul_rn(G, __dadd_rn(m[j], __dmul_rn(__dmul_rn(coef[j], m[j]), fst)));  // as K2 / nbody.cc:14-16,70
}

#ifndef NB_F64L_R
#define NB_F64L_R 2
#endif
#ifndef NB_F64L_BATCH
#define NB_F64L_BATCH 4
#endif
constexpr int F64L_R = NB_F64L_R;          // targets per lane
constexpr int F64L_BATCH = NB_F64L_BATCH;  // sources per scalar-load batch: 4 planes x 8 dwords = 32 SGPRs, two batches live

template <bool SPLIT, bool ACCEL_ONLY, bool SELFCHECK>
__global__ __launch_bounds__(WG, 4) void nbody_force_f64_large(F64LargeArgs a) {
    constexpr int R = F64L_R, U = F64L_BATCH;
    const int t = threadIdx.x, n = a.n;
    const long base = (long)blockIdx.x * (WG * R);
    const double* __restrict__ qx = a.q;
    const double* __restrict__ qy = a.q + n;
    const double* __restrict__ qz = a.q + 2 * (size_t)n;
    const double* __restrict__ gm = a.gm;
    int idx[R];
    double xi[R], yi[R], zi[R], ax[R], ay[R], az[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long i = base + (long)r * WG + t;
        idx[r] = (int)(i < n ? i : n - 1);
        xi[r] = qx[idx[r]]; yi[r] = qy[idx[r]]; zi[r] = qz[idx[r]];
        ax[r] = ay[r] = az[r] = 0.0;
    }
    // slice of the sources for this blockIdx.y, in 256-source tiles
    const int ntiles = (n + TILE - 1) / TILE;
    const int per = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y;
    int j0 = (int)blockIdx.y * per * TILE, j1 = j0 + per * TILE;
    if (j0 > n) j0 = n;
    if (j1 > n) j1 = n;

    auto interact = [&](double sx, double sy, double sz, double sg, int j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double dx = sx - xi[r], dy = sy - yi[r], dz = sz - zi[r];
            double r2 = dx * dx + dy * dy + dz * dz + a.eps2;
            double y = rsqrt_fast(r2);
            double s = sg * y * y * y;
            if (SELFCHECK) s = (j == idx[r]) ? 0.0 : s;  // only needed when eps == 0: otherwise the self pair is s*0 = 0
            ax[r] += s * dx; ay[r] += s * dy; az[r] += s * dz;
        }
    };

    const int jb = j0 + (j1 - j0) / U * U;
    if (jb > j0) {
        double cx[U], cy[U], cz[U], cg[U], nx[U], ny[U], nz[U], ng[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { cx[u] = qx[j0 + u]; cy[u] = qy[j0 + u]; cz[u] = qz[j0 + u]; cg[u] = gm[j0 + u]; }
        for (int j = j0; j < jb; j += U) {
            const int jn = (j + U < jb) ? j + U : j0;  // last prefetch wraps (harmless re-read)
#pragma unroll
            for (int u = 0; u < U; ++u) { nx[u] = qx[jn + u]; ny[u] = qy[jn + u]; nz[u] = qz[jn + u]; ng[u] = gm[jn + u]; }
#pragma unroll
            for (int u = 0; u < U; ++u) interact(cx[u], cy[u], cz[u], cg[u], j + u);
#pragma unroll
            for (int u = 0; u < U; ++u) { cx[u] = nx[u]; cy[u] = ny[u]; cz[u] = nz[u]; cg[u] = ng[u]; }
        }
    }
    for (int j = jb; j < j1; ++j) interact(qx[j], qy[j], qz[j], gm[j], j);

#pragma unroll
    for (int r = 0; r < R; ++r) {
        long i = base + (long)r * WG + t;
        if (i >= n) continue;
        if (SPLIT) {
            double* p = a.partial + (size_t)blockIdx.y * 3 * n;
            p[i] = ax[r]; p[n + i] = ay[r]; p[2 * (size_t)n + i] = az[r];
        } else if (ACCEL_ONLY) {
            a.acc_out[i] = ax[r]; a.acc_out[n + i] = ay[r]; a.acc_out[2 * (size_t)n + i] = az[r];
        } else {  // kick, drift (nbody.cc:76-88), non-contracted like K2
            double vx = __dadd_rn(a.v[i], __dmul_rn(ax[r], a.dt));
            double vy = __dadd_rn(a.v[n + i], __dmul_rn(ay[r], a.dt));
            double vz = __dadd_rn(a.v[2 * (size_t)n + i], __dmul_rn(az[r], a.dt));
            a.v[i] = vx; a.v[n + i] = vy; a.v[2 * (size_t)n + i] = vz;
            a.qout[i] = __dadd_rn(xi[r], __dmul_rn(vx, a.dt));
            a.qout[n + i] = __dadd_rn(yi[r], __dmul_rn(vy, a.dt));
            a.qout[2 * (size_t)n + i] = __dadd_rn(zi[r], __dmul_rn(vz, a.dt));
        }
    }
}

template <bool ACCEL_ONLY>
__global__ __launch_bounds__(WG) void nbody_reduce_update_f64(F64LargeArgs a, int js) {
    const int i = blockIdx.x * WG + threadIdx.x, n = a.n;
    if (i >= n) return;
    double ax = 0, ay = 0, az = 0;
    for (int s = 0; s < js; ++s) {
        const double* p = a.partial + (size_t)s * 3 * n;
        ax += p[i]; ay += p[n + i]; az += p[2 * (size_t)n + i];
    }
    if (ACCEL_ONLY) {
        a.acc_out[i] = ax; a.acc_out[n + i] = ay; a.acc_out[2 * (size_t)n + i] = az;
    } else {
        double vx = __dadd_rn(a.v[i], __dmul_rn(ax, a.dt));
        double vy = __dadd_rn(a.v[n + i], __dmul_rn(ay, a.dt));
        double vz = __dadd_rn(a.v[2 * (size_t)n + i], __dmul_rn(az, a.dt));
        a.v[i] = vx; a.v[n + i] = vy; a.v[2 * (size_t)n + i] = vz;
        a.qout[i] = __dadd_rn(a.q[i], __dmul_rn(vx, a.dt));
        a.qout[n + i] = __dadd_rn(a.q[n + i], __dmul_rn(vy, a.dt));
        a.qout[2 * (size_t)n + i] = __dadd_rn(a.q[2 * (size_t)n + i], __dmul_rn(vz, a.dt));
    }
}

// slices so that the grid has >= 8 workgroups per CU, each slice >= 8 tiles, at most 64
int plan_f64_large_slices(int n, int n_cus) {
    const long bx = (n + WG * F64L_R - 1) / (WG * F64L_R);
    const long ntiles = (n + TILE - 1) / TILE;
    long js = 1;
    while (bx * js < 8L * n_cus && js < 64 && js * 2 * 8 <= ntiles) js <<= 1;
    return (int)js;
}

template <bool ACCEL_ONLY, bool SELFCHECK>
static int launch_large(const F64LargeArgs& a, hipStream_t stream) {
    const int n = a.n, js = a.j_split;
    const unsigned bx = (unsigned)((n + WG * F64L_R - 1) / (WG * F64L_R)), b1 = (unsigned)((n + WG - 1) / WG);
    hipLaunchKernelGGL(nbody_gm_f64, dim3(b1), dim3(WG), 0, stream, a.m, a.coef, a.gm, n, a.fst, a.G);
    if (js <= 1) {
        hipLaunchKernelGGL((nbody_force_f64_large<false, ACCEL_ONLY, SELFCHECK>), dim3(bx), dim3(WG), 0, stream, a);
        return (int)hipGetLastError();
    }
    if (!a.partial) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((nbody_force_f64_large<true, ACCEL_ONLY, SELFCHECK>), dim3(bx, (unsigned)js), dim3(WG), 0, stream, a);
    if (hipError_t e = hipGetLastError()) return (int)e;
    hipLaunchKernelGGL((nbody_reduce_update_f64<ACCEL_ONLY>), dim3(b1), dim3(WG), 0, stream, a, js);
    return (int)hipGetLastError();
}

int launch_f64_large(const F64LargeArgs& a, hipStream_t stream) {
    const bool accel = a.acc_out != nullptr, selfcheck = !(a.eps2 >= F64_EPS2_MIN);
    if (accel) return selfcheck ? launch_large<true, true>(a, stream) : launch_large<true, false>(a, stream);
    return selfcheck ? launch_large<false, true>(a, stream) : launch_large<false, false>(a, stream);
}

template <int S>
static int launch_s(const F64Args& a, hipStream_t stream) {
    constexpr int TPB = K2_WG / S;
    // a monitor-only launch (do_update == 0) still needs every owner lane when a missile-arrival snapshot may be due
    int blocks = (a.do_update || a.snap_q) ? (a.n + TPB - 1) / TPB : 1;
    if (a.eps2 >= F64_EPS2_MIN) hipLaunchKernelGGL((nbody_step_f64<S, false>), dim3(blocks), dim3(K2_WG), 0, stream, a);
    else hipLaunchKernelGGL((nbody_step_f64<S, true>), dim3(blocks), dim3(K2_WG), 0, stream, a);
    return (int)hipGetLastError();
}

template <int S>
static int launch_batched_s(const F64BatchArgs& b, int n, hipStream_t stream) {
    constexpr int TPB = K2_WG / S;
    bool eps_positive = true;
    for (int k = 0; k < b.count; ++k) eps_positive &= (b.item[k].n <= 0 || b.item[k].eps2 >= F64_EPS2_MIN);
    if (eps_positive) hipLaunchKernelGGL((nbody_step_f64_batched<S, false>), dim3((n + TPB - 1) / TPB, b.count), dim3(K2_WG), 0, stream, b);
    else hipLaunchKernelGGL((nbody_step_f64_batched<S, true>), dim3((n + TPB - 1) / TPB, b.count), dim3(K2_WG), 0, stream, b);
    return (int)hipGetLastError();
}

int launch_f64_batched(const F64BatchArgs& b, int n, int S, hipStream_t stream) {
    if (b.count <= 0 || b.count > MAX_BATCH) return (int)hipErrorInvalidValue;
    switch (S) {
        case 1: return launch_batched_s<1>(b, n, stream);
        case 2: return launch_batched_s<2>(b, n, stream);
        case 4: return launch_batched_s<4>(b, n, stream);
        case 8: return launch_batched_s<8>(b, n, stream);
        case 16: return launch_batched_s<16>(b, n, stream);
        case 32: return launch_batched_s<32>(b, n, stream);
        case 64: return launch_batched_s<64>(b, n, stream);
    }
    return (int)hipErrorInvalidValue;
}

__global__ void nbody_ctl_advance(F64CtlBatch b, int by) {
    const int k = threadIdx.x;
    if (k < b.count && b.ctl[k] && b.ctl[k]->active) b.ctl[k]->base_step += by;
}

__global__ __launch_bounds__(WG) void nbody_fst_fill(F64CtlBatch b, int by, int chunk, const double* __restrict__ table,
                                                      int table_len) {
    const int k = blockIdx.y, t = blockIdx.x * WG + threadIdx.x;
    if (k >= b.count || !b.ctl[k] || !b.fst_chunk[k] || t > chunk + 1) return;
    const F64Ctl c = *b.ctl[k];
    int at = c.base_step + (c.active ? by : 0) + t;
    at = at < 0 ? 0 : (at < table_len ? at : table_len - 1);  // past the end of every run: never used
    b.fst_chunk[k][t] = table[at];
}

int launch_fst_fill(const F64CtlBatch& b, int by, int chunk, const double* table, int table_len, hipStream_t stream) {
    if (b.count <= 0 || chunk < 0 || !table || table_len <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(nbody_fst_fill, dim3((unsigned)((chunk + 2 + WG - 1) / WG), (unsigned)b.count), dim3(WG), 0, stream, b, by,
                       chunk, table, table_len);
    return (int)hipGetLastError();
}

int launch_ctl_advance(const F64CtlBatch& b, int by, hipStream_t stream) {
    hipLaunchKernelGGL(nbody_ctl_advance, dim3(1), dim3(64), 0, stream, b, by);
    return (int)hipGetLastError();
}

int launch_f64(const F64Args& a, int S, hipStream_t stream) {
    switch (S) {
        case 1: return launch_s<1>(a, stream);
        case 2: return launch_s<2>(a, stream);
        case 4: return launch_s<4>(a, stream);
        case 8: return launch_s<8>(a, stream);
        case 16: return launch_s<16>(a, stream);
        case 32: return launch_s<32>(a, stream);
        case 64: return launch_s<64>(a, stream);
    }
    return (int)hipErrorInvalidValue;
}

// lanes per target: as many as keep n*S threads within one 256-thread workgroup per CU — S = 64 (a whole wave per target)
// for every testcase size, n = 1024 included (256 workgroups of 4 targets).  The launch is latency-bound, so the shortest
// per-lane pair loop wins: measured at n = 1024 with graph replay, S = 64 5.3 us/step, S = 32 (128 workgroups, 32 pairs per
// lane) 7.1, S = 16 10.5 (profiles/r02_scenario_batch_timing.txt; NB_F64_SPLIT overrides for experiments).
int auto_split_f64(int n, int n_cus) {
    // beyond the testcase sizes the launch is compute-bound instead: give every SIMD ~4 waves of fp64 work
    long want = (long)n_cus * K2_WG * (n > K2_TILE ? 8 : 1);
    // round 5 (bench/f64_mid_n_sweep.py, profiles/r05_f64_mid_n_sweep.txt): between the testcase sizes and K1s-f64's threshold one
    // doubling less is 4-13 % faster (6144 / 8192 bodies: S = 32, not 64; 12288: 16, not 32); from 16384 on the old bound stays the
    // better one for the eps = 0 systems that still come here (28672: S = 16 is 10 % ahead of 8)
    if (n > K2_TILE && n < 16384) want = want * 5 / 8;
    int S = 1;
    while (S < 64 && (long)n * S * 2 <= want) S <<= 1;
    return S;
}

}  // namespace nbk
