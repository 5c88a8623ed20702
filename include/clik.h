/*
 * clik.h - C ABI of the MI355X batched CLIK hot path.
 *
 * Drop-in boundary for the per-tick solve of the reference controllers:
 *   - PseudoInverseController.solve      casclik/controllers/pseudo_inverse.py:512-556
 *     (evaluating the per-mode functions built at :259-483)
 *   - ReactiveQPController.solve         casclik/controllers/reactive_qp.py:461-528
 *     (H/A/lbA/ubA functions built at :175-298, qpOASES call at :491-513)
 *
 * In the reference that boundary is the CasADi-generated JIT C
 * (`int f(const double** arg, double** res, casadi_int* iw, double* w, int mem)`,
 * dense column-major doubles, caller-owned memory) plus `cs.conic`.  Here the
 * symbolic setup is replaced by a flat POD "skill descriptor" and the per-tick
 * call by one stream-ordered batched launch.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions; every entry returns
 *     0 on success and a negative CLIK_E* code otherwise; clik_last_error()
 *     gives the message of the calling thread's last failure.
 *   - All batch buffers are DEVICE pointers, row-major [B][n] fp64 (what
 *     numpy / torch hand out).  The caller owns every buffer; the library owns
 *     only the handle (a device-resident copy of the descriptor).
 *   - `tterms` is a HOST pointer to 2*n_tslots doubles (values then time
 *     derivatives of the skill's time-only sub-expressions at this tick); it is
 *     copied into the kernel arguments, so the call stays graph-capturable.
 *   - Handles are immutable after creation: solve and rollout calls are re-entrant
 *     across streams.  Calls are asynchronous w.r.t. the host (no hidden sync); the
 *     rollouts keep their per-call time-slot records in a stream-ordered allocation
 *     (hipMallocAsync / hipFreeAsync on the caller's stream), not in the handle, and
 *     reach it through a pinned staging slot the library copies `tterms` into before
 *     the call returns (the caller may free or overwrite the host array at once).
 *     Consequence: a tick is graph-capturable; a rollout is graph-capturable only
 *     for a skill without time slots (n_tslots == 0: nothing is staged) - the
 *     time-slot records of a rollout are consumed at call time, not at replay.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream) so this
 *     header needs no HIP include.
 */
#ifndef CLIK_H
#define CLIK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIK_ABI_VERSION 6

#define CLIK_MAX_DOF     14   /* n_state = n_robot_var + n_virtual_var (two 7-DoF arms in one skill, or one arm
                                 with virtual variables; more than 8 only in the kernels instantiated for the
                                 skill and in the global-workspace QP kernels; ABI 6: was 10)                 */
#define CLIK_MAX_JOINTS  12   /* chain joints, fixed ones included                */
#define CLIK_MAX_TASKS   24   /* constraints per skill (more than 8: the built-in dynamic-shape kernels) */
#define CLIK_MAX_M       14   /* rows of one constraint expression; more than
                                 CLIK_DYN_MAX_M only in the shape-specialised kernels
                                 (attached or AOT): the built-in kernels refuse them  */
#define CLIK_DYN_MAX_M    8
#define CLIK_MAX_ROWS   128   /* affine rows over all constraints                 */
#define CLIK_MAX_SETS    10   /* SetConstraints -> 2^10 modes (more than 6 sets: the built-in mode-scan kernels, which
                                 walk the reference's mode table; a 7-DoF arm with one 1-D set per joint has 128, six
                                 joint limits and three walls 512)                 */
#define CLIK_MAX_TSLOTS  32   /* time-only sub-expressions evaluated by the host  */
#define CLIK_MAX_YTERMS   4   /* input_var terms per affine row                   */
#define CLIK_MAX_QPVARS  46   /* n_state + n_slack of the reactive QP (CLIK_MAX_DOF + one slack per row; the
                                 kernels instantiated for a skill fold the slack of soft equalities away,
                                 the built-in ones are bounded by their rows)       */
#define CLIK_MAX_QPROWS  32   /* constraint rows of the reactive QP               */

/* error codes */
#define CLIK_OK            0
#define CLIK_EINVAL       -1  /* malformed descriptor / argument                  */
#define CLIK_EUNSUPPORTED -2  /* valid reference skill outside the device limits  */
#define CLIK_EHIP         -3  /* HIP runtime failure                              */
#define CLIK_ENOMEM       -4

/* joint types (URDF chain:  T_i = Trans(p) * R * Rot(axis, q) | Trans(axis*q)) */
#define CLIK_JOINT_FIXED     0
#define CLIK_JOINT_REVOLUTE  1
#define CLIK_JOINT_PRISMATIC 2

typedef struct clik_joint {
    int32_t type;
    int32_t q_index;          /* index into the state vector, -1 for fixed        */
    double  R[9];             /* origin rotation RPY(rpy), row-major              */
    double  p[3];             /* origin translation xyz                           */
    double  axis[3];          /* unit joint axis in the joint frame               */
} clik_joint;

/* One affine row over the task features
 *      r = a.z + b.p(z) + g.vec(R(z)) + h.o(z,y) + sum_k yc[k]*y[yi[k]] + c + tval[t_slot]
 * z = [robot_var; virtual_var], p / R = tool position / rotation (row-major
 * vec), o = orientation error 1/2 sum_i r_i x r_i,des w.r.t. the skill's
 * quaternion target.  Replaces the user's CasADi expression graph
 * (casclik/constraints.py:21-24) for the affine-in-features family. */
#define CLIK_ROW_HAS_Q 1
#define CLIK_ROW_HAS_P 2
#define CLIK_ROW_HAS_R 4
#define CLIK_ROW_HAS_O 8
#define CLIK_ROW_HAS_Y 16
#define CLIK_ROW_HAS_T 32

typedef struct clik_row {
    double  a[CLIK_MAX_DOF];
    double  b[3];
    double  g[9];
    double  h[3];
    double  c;
    double  yc[CLIK_MAX_YTERMS];
    int32_t yi[CLIK_MAX_YTERMS];
    int32_t n_y;
    int32_t t_slot;           /* -1: none                                         */
    int32_t flags;            /* CLIK_ROW_HAS_*                                   */
    int32_t _pad;
} clik_row;

/* how output row i of a constraint is formed from affine rows */
#define CLIK_OUT_AFFINE 0     /* e_i = r[row0]                                    */
#define CLIK_OUT_NORM2  1     /* e_i = || r[row0 .. row0+nrows) ||_2              */
/* e_i, its state gradient and d e_i/d t come from a device function generated from the
 * caller's expression graph and compiled into the kernel attached with
 * clik_pinv_attach_kernel / clik_qp_attach_kernel (the CasADi-generated function of
 * pseudo_inverse.py:476-483 / reactive_qp.py:262-298).  All outputs of such a constraint
 * are EXTERN; each owns one placeholder row (zero coefficients; HAS_P flags a use of the
 * chain's tool frame).  Without an attached kernel every solve entry returns
 * CLIK_EUNSUPPORTED: the built-in kernels cannot evaluate it. */
#define CLIK_OUT_EXTERN 2

/* constraint classes (casclik/constraints.py:88,148,299,336) */
#define CLIK_CLS_EQ     0
#define CLIK_CLS_SET    1
#define CLIK_CLS_VELEQ  2
#define CLIK_CLS_VELSET 3

#define CLIK_ATTR_GAIN     1
#define CLIK_ATTR_SET_MIN  2
#define CLIK_ATTR_SET_MAX  4
#define CLIK_ATTR_TARGET   8

typedef struct clik_task {
    int32_t cls;
    int32_t m;                          /* expression rows                        */
    int32_t soft;                       /* constraint_type == "soft"              */
    int32_t gain_is_matrix;             /* 0: scalar gain[0]; 1: m x m row-major  */
    int32_t attr_ext;                   /* CLIK_ATTR_* bits: attributes given as expressions of (t, q, x, y)
                                         * (constraints.py:35-39 allows MX gains and bounds); their values
                                         * come from code generated for the skill, per instance and tick   */
    int32_t out_kind[CLIK_MAX_M];
    int32_t out_row0[CLIK_MAX_M];
    int32_t out_nrows[CLIK_MAX_M];
    double  gain[CLIK_MAX_M * CLIK_MAX_M];
    double  set_min[CLIK_MAX_M];
    double  set_max[CLIK_MAX_M];
    double  target[CLIK_MAX_M];         /* VelocityEqualityConstraint.target      */
    double  slack_weight;
} clik_task;

typedef struct clik_skill_desc {
    int32_t abi_version;                /* CLIK_ABI_VERSION                       */
    int32_t n_q;                        /* robot_var size                         */
    int32_t n_x;                        /* virtual_var size (state = [q; x])      */
    int32_t n_y;                        /* input_var size (0: none)               */
    int32_t n_joints;
    int32_t n_tasks;                    /* constraints, already priority-sorted   */
    int32_t n_rows;
    int32_t n_tslots;
    int32_t uses_fk;                    /* any row with P / R / O features        */
    int32_t quat_src;                   /* orientation target: 0 none, 1 const, 2 input */
    int32_t quat_yi[4];                 /* indices (x,y,z,w) into y when quat_src==2 */
    double  quat[4];                    /* (x,y,z,w) when quat_src==1             */
    clik_joint joints[CLIK_MAX_JOINTS];
    clik_task  tasks[CLIK_MAX_TASKS];
    clik_row   rows[CLIK_MAX_ROWS];
} clik_skill_desc;

/* options of PseudoInverseController (pseudo_inverse.py:42-66) */
#define CLIK_PINV_DAMPED   0
#define CLIK_PINV_STANDARD 1
typedef struct clik_pinv_opts {
    int32_t feedforward;                /* default 1                              */
    int32_t multidim_sets;              /* default 0                              */
    int32_t converge_final_set_to_max;  /* default 0                              */
    int32_t pinv_method;                /* default CLIK_PINV_DAMPED               */
    double  damping_factor;             /* default 1e-7                           */
} clik_pinv_opts;

/* cost weights of ReactiveQPController (reactive_qp.py:44,58-133,175-189) */
typedef struct clik_qp_opts {
    double  weight_shifter;             /* mu, default 1e-3                       */
    double  state_weights[CLIK_MAX_DOF];      /* robot then virtual weights       */
    double  slack_weights[CLIK_MAX_QPROWS];   /* one per soft row, in row order   */
    int32_t max_iter;                   /* active-set iteration cap (0: default)  */
    int32_t _pad;
} clik_qp_opts;

typedef struct clik_pinv clik_pinv;     /* opaque handles                         */
typedef struct clik_qp   clik_qp;

const char* clik_last_error(void);
int32_t     clik_abi_version(void);

/* ---- PseudoInverseController path -------------------------------------- */
/* replaces setup_problem_functions() (pseudo_inverse.py:453-483): validates the
 * skill, builds the mode table (:107-130) and uploads the descriptor.        */
int clik_pinv_create(const clik_skill_desc* desc, const clik_pinv_opts* opts,
                     clik_pinv** out);
/* the same handle WITHOUT a device (no allocation, no upload): answers the host-side queries only
 * (clik_pinv_image_words, kernel names, clik_shape_describe) - how casclik_amd/jit.py instantiates kernels ahead of
 * time on a machine without a GPU; every solve / rollout entry point refuses it (CLIK_EINVAL).  The environment
 * variable CLIK_HOST_ONLY=1 makes every create of a process behave like this.                                  */
int clik_pinv_create_host(const clik_skill_desc* desc, const clik_pinv_opts* opts,
                          clik_pinv** out);
int clik_pinv_destroy(clik_pinv* h);
int clik_pinv_n_modes(const clik_pinv* h);
/* name of the kernel variant serving this skill: an AOT shape name (guard-free
 * instantiation for the skill's structure) or "dynamic" (run-time guards).   */
const char* clik_pinv_kernel_name(const clik_pinv* h);
/* which variant of that kernel a batch of B instances gets: "team4" (four lanes per instance,
 * small batches of the priority-stack family), "mp2"/"mp4" (one wave per mode), "split", "lane"
 * (one instance per lane), "lane/occ2" (its large-batch build), "team4v" (team4 with the skill's
 * numbers compiled in) or "dynamic".                                                    */
const char* clik_pinv_kernel_variant(const clik_pinv* h, int64_t B);
/* developer aid (host only, no GPU needed): writes the C++ ShapeDesc initialiser
 * this skill maps to into buf; returns 1 if the skill is eligible for an AOT
 * shape-specialised kernel, 0 if not, <0 on error (tools/gen_shapes.py).      */
int clik_shape_describe(const clik_skill_desc* desc, const clik_pinv_opts* opts, char* buf, int cap);
/* attach a shape-specialised kernel that was instantiated at run time from the
 * library's kernel templates for exactly this skill structure (the analogue of
 * CasADi's JIT at setup_problem_functions, pseudo_inverse.py:476-483).  The
 * function pointers come from a shared object built by casclik_amd/jit.py.
 * rollout_fn may be NULL (the rollout then keeps the kernel chosen at creation). */
int clik_pinv_attach_kernel(clik_pinv* h, void* solve_fn, void* rollout_fn, const char* name);

/* Value-specialised kernel: the four-lanes-per-instance kernel instantiated with THIS skill's numbers (chain
 * constants, gains, bounds, row coefficients) compiled in, as the functions CasADi generates for the reference
 * are (pseudo_inverse.py:476-483).  clik_pinv_image_words returns the numbers as the 64-bit words of the skill
 * image (n words, or a negative error); casclik_amd/jit.py compiles them into the kernel templates and attaches
 * the result (solve_fn: the per-tick kernel; rollout_fn: its on-device rollout, may be NULL).  They serve the
 * batch sizes the "team4" variant serves (reported as "team4v"); solve_fn = NULL detaches both.  Only for skills
 * of that kernel's family.                                                                                  */
int clik_pinv_image_words(const clik_pinv* h, uint64_t* buf, int cap);
int clik_pinv_attach_value_kernel(clik_pinv* h, void* solve_fn, void* rollout_fn);

/* ---- resident ticks (round 3): closed-loop ticks without a launch per tick --------------------------------
 * ONE launch of the value-specialised four-lanes-per-instance kernel stays on the device and runs tick k as soon as
 * the producer of the inputs has published ticket k.  Protocol (clik_ticket lives in DEVICE memory, 256 bytes,
 * zeroed by the caller before the launch):
 *   producer, per tick k = 1, 2, ...: write q / y (device memory), then  in_seq = k  with release semantics
 *     (a device-side atomic store, or a stream-ordered copy of 4 bytes behind the copies of the inputs);
 *   kernel: every wave waits for in_seq >= k, reads its rows (cache-bypassing loads), runs the tick, writes dq / mode
 *     (write-through), and - once those stores are acknowledged - writes k into ITS OWN slot done[w] (when ticket k + 1
 *     was already published the wave goes straight on and publishes done[w] = k up to one tick later); tick k is
 *     complete when every one of the `waves` slots holds k (waves is filled in by the kernel; clik_pinv_resident_waves
 *     returns it beforehand; `done` is device memory, `waves` words, zeroed by the caller);
 *   ring: with ticket->ring_depth = D > 1 the arrays are rings of D slots - q [D][B][n_q], y [D][B][n_y], dq [D][B][n_q],
 *     mode [D][B] - and tick k reads and writes slot (k - 1) % D.  With ONE buffer a producer can only write the next
 *     inputs after every wave has finished the current tick; with D >= 3 it writes tick k + 1's rows (slot k % D, free
 *     once every done[w] >= k + 1 - D) while tick k runs and publishes ticket k + 1 ahead: the kernel then never waits;
 *   state on the device (clik_pinv_resident_run_state): the kernel reads q once (tick 1) and integrates it itself,
 *     q += clamp(dq, +-max_speed) * integrate_dt after every tick (the notebooks' loop, ur5_moe2016_example2.ipynb:537-545),
 *     so only the TARGETS y come from outside: a producer that streams them ahead through the ring never stalls the
 *     kernel, and the loop over q closes without leaving it; dq (clamped) and mode are written per tick as before;
 *   the kernel leaves after n_ticks, when anyone writes stop != 0, or when its watchdog expires (a budget of polls over
 *     its whole life, timeout_s at a nominal 0.2 us per poll; it then writes stop = 2 itself) - it never spins unguarded.
 * clik_ticket_feed launches the reference producer (one device block that publishes tickets 1 .. n_ticks, either
 * back to back or - closed_loop - each only after every slot shows the previous tick) on `stream`, which must not share
 * a HARDWARE QUEUE with the kernel's stream (the runtime multiplexes the streams of one priority onto a few queues; a
 * producer or copy queued behind the resident kernel waits until the watchdog lets it go): use a stream of another
 * priority (hipStreamCreateWithPriority).  Only for handles with an attached value-specialised kernel of that family
 * (clik_pinv_attach_resident_kernel; casclik_amd/jit.py does it).  Not graph-capturable.  Measured on one MI355X
 * (tools/resident_probe.py, profiles/r6_resident_probe.txt; round 6), 16384 instances: 2.6 us per tick when the producer
 * publishes ahead (a ring of four slots), 5.4 us when it waits for done[] (closed loop), 3.66 us for one launch per tick. */
typedef struct clik_ticket {
    uint32_t in_seq;     uint32_t _p0[15];
    uint32_t ring_depth; uint32_t _p1a;      /* input / output slots, set by the caller before the launch (0 or 1: one buffer) */
    double   integrate_dt;                   /* written by clik_pinv_resident_run_state: the kernel keeps the state - it     */
    double   max_speed;                      /*   reads q at tick 1 only and then steps q += clamp(dq, +-max_speed) * dt     */
    uint32_t _p1[10];
    uint32_t stop;       uint32_t _p2[15];
    uint32_t waves;      uint32_t ticks_done;  uint32_t _p3[14];
} clik_ticket;
int clik_pinv_attach_resident_kernel(clik_pinv* h, void* resident_fn);
int clik_pinv_resident_waves(const clik_pinv* h, int64_t B);
int clik_pinv_resident_run(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                           const double* y, double* dq, int32_t* mode, clik_ticket* ticket, uint32_t* done,
                           double timeout_s, void* stream);
/* ... with the state kept by the kernel: q is read at tick 1 and stepped with q += clamp(dq, +-max_speed) * integrate_dt
 * after every tick (max_speed 0: no clamp); the two numbers are written into the ticket by the call, on `stream`.   */
int clik_pinv_resident_run_state(const clik_pinv* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                                 const double* y, double* dq, int32_t* mode, clik_ticket* ticket, uint32_t* done,
                                 double integrate_dt, double max_speed, double timeout_s, void* stream);
int clik_ticket_feed(clik_ticket* ticket, const uint32_t* done, int32_t n_ticks, int32_t closed_loop,
                     int32_t waves_per_tick, double timeout_s, void* stream);

/* Resident ticks of the ReactiveQPController (round 5; the per-tick body of reactive_qp.py:461-528 as ONE launch that
 * solves tick k's QP whenever ticket k is published - same ticket, same `done` slots (clik_qp_resident_waves of them),
 * same ring of input / output slots as clik_pinv_resident_run).  For bound-constrained skills without virtual
 * variables whose value-specialised kernel is attached (clik_qp_attach_resident_kernel, done by casclik_amd.jit);
 * CLIK_EUNSUPPORTED otherwise, or when B needs more waves than the device keeps resident.  Every instance's working
 * set stays in the kernel from tick to tick: tick 1 is a cold solve, every later tick is hot-started, as the
 * reference's qpOASES instance is (reactive_qp.py:491-513).
 *   q [ring][B][n_q], y [ring][B][n_y] (device, read in place)  ->  dq [ring][B][n_q], slack [ring][B][n_slack] or NULL,
 *   status [ring][B] or NULL                                                                                          */
int clik_qp_attach_resident_kernel(clik_qp* h, void* resident_fn);
int clik_qp_resident_waves(const clik_qp* h, int64_t B);
int clik_qp_resident_run(const clik_qp* h, int64_t B, int32_t n_ticks, const double* tterms, const double* q,
                         const double* y, double* dq, double* slack, int32_t* status, clik_ticket* ticket,
                         uint32_t* done, double timeout_s, void* stream);

/* replaces solve() (pseudo_inverse.py:512-556) for B instances at once.
 *   q  [B][n_q]   x [B][n_x] or NULL   y [B][n_y] or NULL      (device, in)
 *   dq [B][n_q]   dx [B][n_x] or NULL                           (device, out)
 *   mode [B] int32 or NULL: accepted mode index, -1 if none (then dq = 0)   */
int clik_pinv_solve_batch(const clik_pinv* h, int64_t B, const double* tterms,
                          const double* q, const double* x, const double* y,
                          double* dq, double* dx, int32_t* mode, void* stream);
/* The same with one time per instance (`t` of 1 or B values): a batch of robots at different
 * phases of their trajectories in one launch.  t_inst is a DEVICE array [B][2 * n_tslots] - row b holds what
 * `tterms` holds, evaluated at instance b's time.  Needs a shape-specialised kernel for the skill
 * (CLIK_EUNSUPPORTED otherwise: group the instances by time and call clik_pinv_solve_batch per group, as
 * casclik_amd's controllers do).  A skill without time slots ignores t_inst.                               */
int clik_pinv_solve_batch_t(const clik_pinv* h, int64_t B, const double* t_inst,
                            const double* q, const double* x, const double* y,
                            double* dq, double* dx, int32_t* mode, void* stream);

/* "next" row: n_ticks of solve -> clamp(+-max_speed) ->
 * explicit Euler q += dq*dt inside one launch, the loop every notebook runs
 * on the host (ur5_moe2016_example2.ipynb:537-545).  tterms holds
 * n_ticks*2*n_tslots doubles (host).  q is updated in place; dq/mode receive
 * the last tick.  max_speed <= 0 disables the clamp.                        */
int clik_pinv_rollout_batch(const clik_pinv* h, int64_t B, int32_t n_ticks,
                            double dt, double max_speed, const double* tterms,
                            double* q, const double* y, double* dq,
                            int32_t* mode, void* stream);
/* The same for skills with virtual variables (path following: the loop of
 * cart_on_track_1D_comparison_of_controllers.ipynb cell 60 integrates the path
 * parameter next to the robot state): x [B][n_x] is updated in place like q,
 * dx receives the last tick; the clamp applies to the robot velocities only.
 * Needs a shape-specialised kernel (attached or AOT).                          */
int clik_pinv_rollout_batch_x(const clik_pinv* h, int64_t B, int32_t n_ticks,
                              double dt, double max_speed, const double* tterms,
                              double* q, double* x, const double* y, double* dq,
                              double* dx, int32_t* mode, void* stream);

/* The same with a choice of integration scheme (casclik/integration_methods.py:11-23):
 * CLIK_INTEGRATE_EULER = the calls above; CLIK_INTEGRATE_RK4 = classical Runge-Kutta with the controller
 * as the right-hand side, four solves per tick at t, t+dt/2, t+dt/2, t+dt (each clamped), state
 * += dt/6 (k1 + 2 k2 + 2 k3 + k4).  tterms then holds n_ticks*4*2*n_tslots doubles (one record per stage);
 * dq / dx receive the last tick's combined rate, mode the mode of its first stage.  x / dx may be NULL for
 * skills without virtual variables.  Runge-Kutta needs a shape-specialised kernel.                  */
#define CLIK_INTEGRATE_EULER 0
#define CLIK_INTEGRATE_RK4   1
int clik_pinv_rollout_batch_m(const clik_pinv* h, int64_t B, int32_t n_ticks, int32_t method,
                              double dt, double max_speed, const double* tterms,
                              double* q, double* x, const double* y, double* dq,
                              double* dx, int32_t* mode, void* stream);

/* ---- ReactiveQPController path ----------------------------------------- */
/* replaces setup_problem_functions()+setup_solver() (reactive_qp.py:248-298) */
int clik_qp_create(const clik_skill_desc* desc, const clik_qp_opts* opts,
                   clik_qp** out);
int clik_qp_create_host(const clik_skill_desc* desc, const clik_qp_opts* opts,
                        clik_qp** out);   /* see clik_pinv_create_host */
int clik_qp_destroy(clik_qp* h);
int clik_qp_n_vars(const clik_qp* h);   /* n_state + n_slack                     */
int clik_qp_n_rows(const clik_qp* h);
/* device memory the handle holds as work area of the kernels that keep theirs in global memory (QPs beyond 16 rows or
 * eight states on the built-in kernels; 0 for every other skill): sized for the 64-instance blocks the largest batch
 * so far needed, at most the device's resident blocks; grown by retiring the smaller area (captured graphs stay
 * valid); released by clik_qp_destroy.  See INTEGRATION.md. */
int64_t clik_qp_workspace_bytes(const clik_qp* h);
/* kernel serving the skill: an AOT shape name, "jit_<hash>" or "dynamic"       */
const char* clik_qp_kernel_name(const clik_qp* h);
/* as clik_shape_describe / clik_pinv_attach_kernel, for the QP controller: the
 * shape-specialised QP kernel (soft equalities eliminated, active set over the
 * remaining rows) is instantiated per skill structure; reactive_qp.py:283-298
 * JIT-compiles its H/A/lbA/ubA functions at the same point.                   */
int clik_qp_shape_describe(const clik_skill_desc* desc, char* buf, int cap);
int clik_qp_attach_kernel(clik_qp* h, void* solve_fn, void* rollout_fn, const char* name);
/* Value-specialised per-tick QP kernel (see clik_pinv_attach_value_kernel): clik_qp_image_words returns the skill
 * image and the QP options (weights, weight shifter, iteration cap) of the handle as 64-bit words;
 * casclik_amd/jit.py compiles them into the kernel template and attaches the result, which then serves
 * clik_qp_solve_batch / _hot at every batch size (solve_fn = NULL detaches); rollout_fn (may be NULL) is the
 * on-device rollout of the same instantiation, used for the bound-constrained family.  Needs a shape-specialised
 * kernel.                                                                                                       */
int clik_qp_image_words(const clik_qp* h, uint64_t* buf, int cap);
int clik_qp_attach_value_kernel(clik_qp* h, void* solve_fn, void* rollout_fn);
/* 1 when the skill's QP, after the soft equalities are eliminated, is bound-constrained (every remaining row a hard
 * bound on one state: joint limits / speed limits) and a shape-specialised kernel serves it - the family whose
 * value-specialised kernel runs without LDS (casclik_amd attaches it by default).                              */
int clik_qp_is_box_family(const clik_qp* h);

/* replaces solve() (reactive_qp.py:461-528).
 *   dq [B][n_q], dx [B][n_x] or NULL, slack [B][n_slack] or NULL  (device, out)
 *   status [B] int32 or NULL: 0 optimal, 1 iteration cap, 2 infeasible       */
int clik_qp_solve_batch(const clik_qp* h, int64_t B, const double* tterms,
                        const double* q, const double* x, const double* y,
                        double* dq, double* dx, double* slack, int32_t* status,
                        void* stream);

/* The same with a hot start (the reference's qpOASES instance hot-starts from the previous
 * call, reactive_qp.py:491-513): hot_set [B] int32 (device, in/out) carries each instance's
 * working set from tick to tick.  use_hot = 0: cold guess, the final set is only written (first
 * tick); use_hot = 1: the stored set seeds the active-set iteration, then is overwritten.  The
 * set is a guess that the solver repairs; the result is the same minimiser.  Only the
 * shape-specialised kernels use it (the dynamic fallback ignores it and leaves it unchanged).  */
int clik_qp_solve_batch_hot(const clik_qp* h, int64_t B, const double* tterms,
                            const double* q, const double* x, const double* y,
                            double* dq, double* dx, double* slack, int32_t* status,
                            int32_t* hot_set, int32_t use_hot, void* stream);
/* ... with one time per instance: t_inst as for clik_pinv_solve_batch_t (hot_set may be NULL).              */
int clik_qp_solve_batch_t(const clik_qp* h, int64_t B, const double* t_inst,
                          const double* q, const double* x, const double* y,
                          double* dq, double* dx, double* slack, int32_t* status,
                          int32_t* hot_set, int32_t use_hot, void* stream);

/* "next" row for the QP controller: n_ticks of solve -> clamp(+-max_speed) ->
 * explicit Euler q += dq*dt inside one launch, the working set hot-started from tick to tick.
 * q is updated in place; dq / slack receive the last tick, status the worst status met (an
 * infeasible tick leaves q where it is).  Needs a shape-specialised kernel for the skill.     */
int clik_qp_rollout_batch(const clik_qp* h, int64_t B, int32_t n_ticks, double dt,
                          double max_speed, const double* tterms, double* q, const double* y,
                          double* dq, double* slack, int32_t* status, void* stream);
/* ... and for skills with virtual variables (see clik_pinv_rollout_batch_x).      */
int clik_qp_rollout_batch_x(const clik_qp* h, int64_t B, int32_t n_ticks,
                            double dt, double max_speed, const double* tterms,
                            double* q, double* x, const double* y, double* dq,
                            double* dx, double* slack, int32_t* status, void* stream);
/* ... with the integration scheme selectable (CLIK_INTEGRATE_EULER / CLIK_INTEGRATE_RK4, see
 * clik_pinv_rollout_batch_m): Runge-Kutta solves four QPs per tick, each stage hot-started from the previous
 * one; tterms then holds n_ticks*4*2*n_tslots doubles.  A tick with an infeasible stage leaves q where it was. */
int clik_qp_rollout_batch_m(const clik_qp* h, int64_t B, int32_t n_ticks, int32_t method, double dt,
                            double max_speed, const double* tterms, double* q, double* x, const double* y,
                            double* dq, double* dx, double* slack, int32_t* status, void* stream);

/* QP data only (H diag, A, lbA, ubA as the reference's H_func/A_func/Blb/Bub,
 * reactive_qp.py:283-298) for inspection and parity tests:
 *   Hdiag [B][nv], A [B][nc][nv] row-major, lbA [B][nc], ubA [B][nc]          */
int clik_qp_data_batch(const clik_qp* h, int64_t B, const double* tterms,
                       const double* q, const double* x, const double* y,
                       double* Hdiag, double* A, double* lbA, double* ubA,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIK_H */
