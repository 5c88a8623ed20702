// Resident ticks of the ReactiveQPController's bound-constrained family: the kernel and its launcher.  Included by
// clik_qp_static.hpp (needs qp_tick_static, QpLayout, QpImg and the ticket protocol of clik_pinv_team.hpp); its own
// file only to keep that header readable.
#pragma once

namespace clik {

// ... RESIDENT ticks of the bound-constrained family (round 5): ONE launch that solves tick k's QP whenever ticket k is
// published (the protocol, the watchdog and the ring of input / output slots of pinv_resident_team_kernel,
// clik_pinv_team.hpp; reference: the per-tick body of ReactiveQPController.solve, reactive_qp.py:461-528, called from a
// loop that feeds it fresh targets).  Four lanes per instance, lane r evaluating the sines / cosines of state variables 2r and
// 2r + 1 (DPP exchange, as pinv_solve_static_values_quad_kernel) - in a resident kernel nobody pays for launching the
// waves, so the 266 instructions the quad saves on the sin / cos evaluations count in full - and
// every instance's working set stays in a register from tick to tick: every tick after the first is hot-started, as the
// reference's qpOASES instance is (reactive_qp.py:491-513).  Lane r of a quad requests elements 2r, 2r + 1 of its
// instance's robot_var / input_var rows and stores the same elements of the velocity and slack rows; lane 0 the status.
template <const ShapeDesc& SD>
constexpr bool qp_resident_ok() { return qp_front4_ok<SD>() && SD.n_x == 0; }
template <const ShapeDesc& SD, class IMGV>
__global__ __launch_bounds__(WAVE) void qp_resident_box_front4_kernel(
    const double* q, const double* y, double* dq, double* slack_out, int32_t* status_out, const long long B,
    const TickArgs tk, ResidentTicket* ticket, unsigned* done, const int n_ticks, const unsigned long long max_polls)
{
    using LY = QpLayout<SD>;
    static_assert(LY::BOX && SD.n_x == 0, "resident QP ticks: box family, robot variables only");
    constexpr int N = SD.n, NY = SD.n_y > 0 ? SD.n_y : 0, NS = LY::NS;
    constexpr QpImg<SD> kValues = IMGV::value;
    const int tid = threadIdx.x;
    const int r = tid & 3;
    const long long inst = (long long)blockIdx.x * (WAVE / 4) + (tid >> 2);
    const bool valid = inst < B;
    const long long binst = valid ? inst : (B - 1);
    ResidentWave rw;
    rw.init(ticket, done, max_polls, n_ticks, blockIdx.x, gridDim.x, tid);
    bool have_next = false;
    const long long ring = rw.ring_depth();
    static_assert(N <= 8, "resident QP kernel: at most eight state variables");
    constexpr int RY = NY > 0 ? (NY + 7) / 8 : 1;
    // (a lane's share of the rows is two doubles each: the NEXT tick's shares are requested before this tick's
    // arithmetic whenever their ticket is already out, and arrive under it)
    double zp[2], yp[2 * RY], zp_next[2], yp_next[2 * RY];
#pragma unroll
    for (int i = 0; i < 2; ++i) zp[i] = zp_next[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 2 * RY; ++i) yp[i] = yp_next[i] = 0.0;
    auto request_rows = [&](const int k, double (&zq)[2], double (&yq)[2 * RY]) __attribute__((always_inline)) {
        const long long row = ((long long)((k - 1) % (int)ring)) * B + binst;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = 2 * r + i;
            zq[i] = __hip_atomic_load(q + row * N + (e < N ? e : N - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if constexpr (NY > 0) {
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) {
                const int e = 8 * (i / 2) + 2 * r + (i & 1);
                yq[i] = __hip_atomic_load(y + row * NY + (e < NY ? e : NY - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };
    int32_t hot_word = 0;       // the instance's working set: in a register for the whole run
    int owed = 0;
    const SinCosK sck = sincos_consts();        // (once per launch)
#pragma unroll 1
    for (int k = 1; k <= n_ticks; ++k) {
        if (!have_next) {
            rw.poll_for((unsigned)k);
            if (rw.leave) break;
            asm volatile("" ::: "memory");
            request_rows(k, zp, yp);
            // (waited for inside this branch: left pending, the compiler - whose wait counts are merged over both ways
            // into the tick - made the fed-ahead path wait for its freshly requested NEXT rows as well)
#pragma unroll
            for (int i = 0; i < 2; ++i) pin_arrived(zp[i]);
            if constexpr (NY > 0) {
#pragma unroll
                for (int i = 0; i < 2 * RY; ++i) pin_arrived(yp[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) zp[i] = zp_next[i];
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) yp[i] = yp_next[i];
        }
        have_next = false;
        if (k < n_ticks && rw.seen >= (unsigned)(k + 1)) {
            request_rows(k + 1, zp_next, yp_next);
            have_next = true;
        }
        // (the ticket word for the next tick's decision is requested NOW - a whole tick to arrive - not at the end of the
        // tick, a few instructions before it is read: see pinv_resident_quad_kernel)
        if (have_next) rw.peek();
        double sn0, cs0, sn1, cs1;
        sincos_fast(zp[0], sn0, cs0, sck);
        sincos_fast(zp[1], sn1, cs1, sck);
        const bool huge = (fabs(zp[0]) > kSinCosFastMax) | (fabs(zp[1]) > kSinCosFastMax);
        if (__builtin_expect(__ballot(huge) != 0ull, 0)) {
            if (fabs(zp[0]) > kSinCosFastMax) { const SinCos sc = sincos_slow(zp[0]); sn0 = sc.s; cs0 = sc.c; }
            if (fabs(zp[1]) > kSinCosFastMax) { const SinCos sc = sincos_slow(zp[1]); sn1 = sc.s; cs1 = sc.c; }
        }
        double z[N], sns[N], css[N], yrow[NY > 0 ? NY : 1];
        static_for<0, N>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int CTRL = (j / 2) * 0x55;
            z[j] = quad_perm_f64<CTRL>(zp[j & 1]);
            if constexpr (shape_state_type(SD, j) == CLIK_JOINT_REVOLUTE) {
                sns[j] = quad_perm_f64<CTRL>((j & 1) ? sn1 : sn0);
                css[j] = quad_perm_f64<CTRL>((j & 1) ? cs1 : cs0);
            } else {
                sns[j] = css[j] = 0.0;
            }
        });
        if constexpr (NY > 0) {
            static_for<0, NY>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int CTRL = ((j % 8) / 2) * 0x55;
                yrow[j] = quad_perm_f64<CTRL>(yp[2 * (j / 8) + (j & 1)]);
            });
        }
        double priv[LY::SLOTS];
        double v[N], sl[LY::NSA];
        // (two copies of the tick, `use_hot` a literal in each - qp_box_values_body, clik_qp_static.hpp: tick 1 is the cold one)
        auto solve = [&](auto hot_c) __attribute__((always_inline)) {
            return qp_tick_static<SD, 1, false, true>(&kValues.img, &kValues.tail, tk, z, yrow, tid & (WAVE - 1), valid, priv, v,
                                                      sl, &hot_word, decltype(hot_c)::value, nullptr, nullptr, sns, css);
        };
        const int status = (k > 1) ? solve(std::true_type{}) : solve(std::false_type{});
        // (the next tick's rows and the ticket word, requested at the top of this tick, are waited for HERE - before this
        // tick's stores are issued - not at the top of the next tick, where the same wait would also cover those stores:
        // pin_arrived, clik_device.hpp.  Unconditional: the compiler's wait insertion is not path-sensitive, and with
        // nothing requested there is nothing to wait for.  The config-3 kernel takes its rows through LDS instead
        // (stage_rows): measured here too, round 6 - 5.76 against 5.55 us per tick: this kernel's tick is long enough for
        // the loads to arrive before the compiler's AGPR copy of their registers asks for them, and the copies' address
        // arithmetic came on top)
#pragma unroll
        for (int i = 0; i < 2; ++i) pin_arrived(zp_next[i]);
        if constexpr (NY > 0) {
#pragma unroll
            for (int i = 0; i < 2 * RY; ++i) pin_arrived(yp_next[i]);
        }
        pin_arrived(rw.seen);
        // (a producer that runs exactly ONE tick ahead: the early look was too early - look again, and wait for it inside
        // the branch)
        if (have_next && k + 1 < n_ticks && rw.seen < (unsigned)(k + 2)) {
            rw.peek();
            pin_arrived(rw.seen);
        }
        if (owed != 0) {
            rw.publish_done(owed);
            owed = 0;
        }
        if (valid) {
            const unsigned bad = (status == 2) ? 0x7ff80000u : 0u;      // (nan_or: the NaN of an infeasible instance, as bits)
            const long long orow = ((long long)((k - 1) % (int)ring)) * B + inst;
            double s0 = v[N - 1], s1 = v[N - 1];
            static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
                constexpr int kk = decltype(kc)::value;
                if constexpr (2 * kk < N) s0 = (r == kk) ? v[2 * kk] : s0;
                if constexpr (2 * kk + 1 < N) s1 = (r == kk) ? v[2 * kk + 1] : s1;
            });
            if (2 * r < N) __hip_atomic_store(dq + orow * N + 2 * r, nan_or(s0, bad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (2 * r + 1 < N) __hip_atomic_store(dq + orow * N + 2 * r + 1, nan_or(s1, bad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if constexpr (NS > 0) {
                if (slack_out != nullptr) {
                    static_for<0, (NS + 7) / 8>([&](auto rc) __attribute__((always_inline)) {
                        constexpr int rho = decltype(rc)::value;
                        double t0 = sl[NS - 1], t1 = sl[NS - 1];
                        static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
                            constexpr int kk = decltype(kc)::value;
                            constexpr int j0 = 8 * rho + 2 * kk;
                            if constexpr (j0 < NS) t0 = (r == kk) ? sl[j0] : t0;
                            if constexpr (j0 + 1 < NS) t1 = (r == kk) ? sl[j0 + 1] : t1;
                        });
                        const int e = 8 * rho + 2 * r;
                        if (e < NS) __hip_atomic_store(slack_out + orow * NS + e, nan_or(t0, bad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (e + 1 < NS) __hip_atomic_store(slack_out + orow * NS + e + 1, nan_or(t1, bad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    });
                }
            }
            if (status_out != nullptr && r == 0)
                __hip_atomic_store(status_out + orow, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // (the next ticket may be out already: then this tick's "done" is published behind the next tick's arithmetic,
        // its stores a whole tick old; otherwise - a closed loop waits for it - at once)
        if (have_next) owed = k;
        else rw.publish_done(k);
    }
    if (owed != 0) rw.publish_done(owed);
}

template <const ShapeDesc& SD, class IMGV>
inline hipError_t launch_qp_resident_values(const TickArgs& tk, long long B, const double* q, const double* y, double* dq,
                                            double* slack, int32_t* status, void* ticket, unsigned* done, int n_ticks,
                                            unsigned long long budget, hipStream_t stream)
{
    if constexpr (qp_resident_ok<SD>()) {
        const unsigned grid = (unsigned)((B + 15) / 16);
        int dev = 0, cus = 0, per_cu = 0;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qp_resident_box_front4_kernel<SD, IMGV>, WAVE, 0);
        if (oe == hipSuccess) oe = hipGetDevice(&dev);
        if (oe == hipSuccess) oe = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (oe != hipSuccess) return oe;
        // every wave resident at once, and room for the ticket feeder: this kernel takes a SIMD's whole register file
        // (one wave per SIMD); keep one CU's worth of SIMDs free
        const long long max_blocks = (long long)(cus - 1) * (per_cu < 4 ? per_cu : 4);
        if ((long long)grid > max_blocks) return hipErrorNotSupported;
        hipLaunchKernelGGL((qp_resident_box_front4_kernel<SD, IMGV>), dim3(grid), dim3(WAVE), 0, stream, q, y, dq, slack,
                           status, B, tk, (ResidentTicket*)ticket, done, n_ticks, budget);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

}  // namespace clik
