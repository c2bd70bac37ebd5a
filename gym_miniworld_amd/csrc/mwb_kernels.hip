// mwb_kernels.hip - HIP kernels of the batched MiniWorld stepper + renderer (gfx950 / MI355X).
//
// Pipeline of one mwb_step():
//   step_kernel   1 thread / env   f64   MiniWorldEnv.step (miniworld.py:658-716) + task rule
//   reset_kernel  1 wave   / env   f64   MiniWorldEnv.reset (miniworld.py:532-592), only envs that ended
//   prep_kernel   1 thread / env   f64->f32  camera basis, per-face lighting, box frame
//   render_kernel 1 workgroup / env f32  render_obs/_render_world (miniworld.py:1059-1085,1160-1220):
//                 room table staged in LDS, one wave per 16x4 pixel tile, 8 coverage rays per pixel
//                 through the portal graph, per-surface shading, framebuffer assembled in LDS and
//                 written with 16-byte coalesced stores.
// No MFMA: there is no dense contraction on this path; the work is rays x rooms (VALU + LDS).
//
// Compiled with -ffp-contract=off: world generation and the step must reproduce the reference's
// float64 results bit for bit, and the visibility arithmetic of the renderer is specified with
// explicit fmaf() ("render spec").
#include "mwb_internal.h"
#include "mwb_glibc_trig.h"

#define WAVE 64

// float64 sin / cos as the reference's process computes them (math.sin / math.cos -> glibc 2.35 __sin_fma / __cos_fma,
// restated bit for bit in mwb_glibc_trig.h); OCML's differ in the last bit for a few arguments in a thousand, which an
// accumulating pose would carry along.  Beyond |x| = 105414350 (unreachable for a heading) the platform's functions.
__device__ __forceinline__ double ref_sin(double x) { return mwb_trig::in_range(x) ? mwb_trig::sin_glibc(x) : sin(x); }
__device__ __forceinline__ double ref_cos(double x) { return mwb_trig::in_range(x) ? mwb_trig::cos_glibc(x) : cos(x); }

// =========================================================================== MT19937 (numpy legacy)
// numpy.random.RandomState draw recipes used through reference random.py:4-65.
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// single-thread generator over a state in global memory (step kernel, domain randomisation draws)
struct MtSerial {
    uint32_t *key;
    int pos;
    __device__ void load(uint32_t *st) { key = st; pos = (int)st[624]; }
    __device__ void store() { key[624] = (uint32_t)pos; }
    __device__ __forceinline__ uint32_t next32() {
        if (pos == 624) {
            for (int i = 0; i < 624; i++) key[i] = mt_mix(key[i], key[(i + 1) % 624], key[(i + 397) % 624]);
            pos = 0;
        }
        return mt_temper(key[pos++]);
    }
    __device__ __forceinline__ double next_double() {
        uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double uniform(double lo, double hi) { double sc = hi - lo; return lo + sc * next_double(); }
};

// wave-cooperative generator over a state staged in LDS (reset kernel): every lane carries the same
// position and computes the same draw; the twist is done by all 64 lanes.
// Pointers into LDS carry their address space in the type: the generator's state is reached through `this` in functions the compiler does
// not inline, where a generic pointer would make every access a FLAT instruction (slower than ds_read, and counted as VMEM).
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS double lds_f64;
typedef LDS_AS int lds_i32;
typedef LDS_AS uint32_t lds_u32;
struct MtWave {
    lds_u32 *key;   // LDS, 624 words
    int pos, lane;
    __device__ __forceinline__ void twist() {
        for (int base = 0; base < 624; base += WAVE) {
            int i = base + lane;
            uint32_t v = 0;
            if (i < 624) v = mt_mix(key[i], key[(i + 1) % 624], key[(i + 397) % 624]);
            __syncthreads();
            if (i < 624) key[i] = v;
            __syncthreads();
        }
        pos = 0;
    }
    __device__ __forceinline__ uint32_t next32() {
        if (pos == 624) twist();
        return mt_temper(key[pos++]);
    }
    __device__ __forceinline__ double next_double() {
        uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double uniform(double lo, double hi) { double sc = hi - lo; return lo + sc * next_double(); }
    // RandomState.randint(lo, hi): masked rejection on 32-bit words, no draw when the range is 1
    __device__ __forceinline__ int randint(int lo, int hi) {
        uint32_t rng = (uint32_t)(hi - 1 - lo);
        if (rng == 0) return lo;
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next32() & mask; } while (v > rng);
        return lo + (int)v;
    }
};

// ================================================================================== step kernel
// circle vs wall segments, reference math.py:25-57 (strict <, y flattened)
template <class SegPtr>
__device__ __forceinline__ bool seg_hit(SegPtr sg, double px, double pz, double radius) {
    double ax = sg[0], az = sg[1], bx = sg[2], bz = sg[3];
    double abx = bx - ax, abz = bz - az;
    double apx = px - ax, apz = pz - az;
    double dotAPAB = (apx * abx + 0.0) + apz * abz;
    double dotABAB = (abx * abx + 0.0) + abz * abz;
    double proj = dotAPAB / dotABAB;
    proj = proj < 0 ? 0 : (proj > 1 ? 1 : proj);
    double cx = ax + proj * abx, cz = az + proj * abz;
    double ddx = cx - px, ddz = cz - pz;
    double dist = sqrt((ddx * ddx + 0.0) + ddz * ddz);
    return dist < radius;
}


// a + b as the reference's interpreter evaluates it when either may be a numpy float32 SCALAR (a MeshEnt's radius under
// NumPy >= 2, entity.py:118-127) and the other a Python float: the Python float is cast to float32, the sum is a float32
__device__ __forceinline__ double tagged_add(double a, bool a_f32, double b, bool b_f32, bool &res_f32) {
    res_f32 = a_f32 || b_f32;
    return res_f32 ? (double)((float)a + (float)b) : a + b;
}
__device__ __forceinline__ double tagged_add(double a, bool a_f32, double b, bool b_f32) { bool f; return tagged_add(a, a_f32, b, b_f32, f); }

__device__ __forceinline__ double box_radius(double size) {
    double sx = size, sz = size;   // entity.py:367-378, Box(size=s): radius = sqrt(sx^2 + sz^2) / 2
    return sqrt(sx * sx + sz * sz) / 2;
}

// One block = 64 environments (lane = env, so every SoA access is a coalesced row) x STEP_PARTS waves.
// Wave 0 carries the sequential part of MiniWorldEnv.step; the circle-vs-segments test of a forward move -
// the only loop, up to 256 segments in Maze - is split over all waves (wave p takes the 8-segment batches
// p, p + STEP_PARTS, ...), which turns one long dependent load chain per env into STEP_PARTS short ones.
#define STEP_PARTS 8
// NBX = the task's number of boxes (1, 2 or 6): a template parameter so that the one-box tasks do not carry six-wide
// select chains through the sequential part
template <int NBX>
__global__ void __launch_bounds__(64 * STEP_PARTS) step_kernel(MwbDev d, const int32_t *__restrict__ actions,
                                                                const uint8_t *__restrict__ skip) {
    __shared__ double s_nx[64], s_nz[64];
    __shared__ int s_move[64], s_hit[64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool in_range = e < d.N;
    const bool dummy = in_range && skip && skip[e];   // the fork's 'dummy' command, vec_env/subproc_vec_env.py:26-31
    const bool live = in_range && !dummy;
    int sc = 0, a = -1;
    double ax = 0, az = 0, adir = 0, nx = 0, nz = 0, turn_step = 0;
    constexpr int NB = NBX;   // entity order: the boxes (red; red, blue / yellow; PutNext's six), then the agent
    double bx[NBX], bz[NBX], by[NBX], brad[NBX], bsz[NBX];
#pragma unroll
    for (int b = 0; b < NBX; b++) { bx[b] = 0; bz[b] = 0; by[b] = 0; brad[b] = 0; bsz[b] = 0; }
    int carried = -1;   // agent.carrying as a box index
    const double arad = d.agent_radius;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.order_state[1]) {   // adopt the dispatch order completed beside the last pass
        d.order_state[0] ^= 1; d.order_state[1] = 0;                  // (nothing reads the maps while step_kernel runs)
    }
    if (part == 0) {
        s_move[lane] = 0; s_hit[lane] = 0;
        if (dummy) {
            d.reward[e] = -99.0f; d.reward64[e] = -99.0; d.done[e] = 0; d.ep_steps[e] = d.step_count[e];
            d.reset_set[e] = 0;
            d.feature[e * 2] = 0.0f; d.feature[e * 2 + 1] = 0.0f;   // info = {"feature": [0, 0]}
        }
        if (live) {
            a = actions[(size_t)e * d.act_stride];   // stride 2: the low words of an int64 action tensor (mwb_step_i64)
            // an int64 value outside int32 is no action of the enum: the reference's `if action == ...` chain ignores it
            // (miniworld.py:670-695), so must this - not alias 2^32 + 2 to move_forward
            if (d.act_stride == 2 && actions[(size_t)e * 2 + 1] != (a >> 31)) a = -1;
            ax = d.agent_x[e]; az = d.agent_z[e]; adir = d.agent_dir[e];
#pragma unroll
            for (int b = 0; b < NBX; b++)
                if (b < NB) {
                    const size_t be = (size_t)b * d.N + e;
                    bx[b] = d.box_x[be]; bz[b] = d.box_z[be]; by[b] = d.box_y[be]; bsz[b] = d.box_size[be]; brad[b] = box_radius(bsz[b]);
                }
            carried = d.carrying[e];
            MtSerial g;
            const bool use_rng = d.domain_rand || d.task == MWB_TASK_SIM2REAL_PUSH;
            if (use_rng) g.load(d.rng + (size_t)e * MWB_MT_WORDS);
            if (NBX == 2 && d.task == MWB_TASK_SIM2REAL_PUSH && a == 2) {
                // "Very crude approximation [of] the physics of box pushing", simtorealpush.py:109-125, before
                // MiniWorldEnv.step: a box the nominal forward move would touch is shoved away by the vector
                // from that position to the box, unless it would then hit a wall, the other box or the agent,
                // and turns by a random angle.  The rink has four wall segments: tested here, in wave 0.
                const double fwd_dist = d.params[MWB_P_FORWARD_STEP].hi[0];
                const double c = ref_cos(adir), s = ref_sin(adir);
                const double npx = ax + c * fwd_dist, npz = az + (-s) * fwd_dist;
                const int ns = d.n_segs[e];
                double wq[4][4];   // the rink's four wall segments, loaded together (16 coalesced loads in flight)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) wq[i][cc] = d.segs[(size_t)((i < ns ? i : 0) * 4 + cc) * d.N + e];
                for (int b = 0; b < 2; b++) {
                    const double vx = bx[b] - npx, vz = bz[b] - npz;
                    const double dist = sqrt((vx * vx + 0.0) + vz * vz);
                    if (dist < arad + brad[b]) {
                        const double qx = bx[b] + vx, qz = bz[b] + vz;
                        bool hit = false;
#pragma unroll
                        for (int i = 0; i < 4; i++) hit = hit || (i < ns && seg_hit(wq[i], qx, qz, brad[b]));
                        if (!hit) {   // entities in list order: box 0, box 1, agent (itself skipped)
                            const int o = 1 - b;
                            double ddx = bx[o] - qx, ddz = bz[o] - qz;
                            hit = sqrt(ddx * ddx + 0.0 + ddz * ddz) < brad[b] + brad[o];
                            if (!hit) { ddx = ax - qx; ddz = az - qz; hit = sqrt(ddx * ddx + 0.0 + ddz * ddz) < brad[b] + arad; }
                        }
                        if (!hit) {
                            bx[b] = qx; bz[b] = qz;
                            const size_t be = (size_t)b * d.N + e;
                            d.box_x[be] = qx; d.box_z[be] = qz;
                            d.box_dir[be] += g.uniform(-3.141592653589793 / 5, 3.141592653589793 / 5);
                        }
                    }
                }
            }
            sc = d.step_count[e] + 1;   // miniworld.py:663
            d.step_count[e] = sc;
            double fwd_step = d.params[MWB_P_FORWARD_STEP].def[0];
            double fwd_drift = d.params[MWB_P_FORWARD_DRIFT].def[0];
            turn_step = d.params[MWB_P_TURN_STEP].def[0];
            if (d.domain_rand) {   // miniworld.py:665-668: three draws every step, whatever the action
                fwd_step = g.uniform(d.params[MWB_P_FORWARD_STEP].lo[0], d.params[MWB_P_FORWARD_STEP].hi[0]);
                fwd_drift = g.uniform(d.params[MWB_P_FORWARD_DRIFT].lo[0], d.params[MWB_P_FORWARD_DRIFT].hi[0]);
                turn_step = g.uniform(d.params[MWB_P_TURN_STEP].lo[0], d.params[MWB_P_TURN_STEP].hi[0]);
            }
            if (use_rng) g.store();
            if (a == 2 || a == 3) {   // move_agent, miniworld.py:608-633
                double fd = (a == 2) ? fwd_step : -fwd_step;
                double c = ref_cos(adir), s = ref_sin(adir);
                // pos + dir_vec*fwd + right_vec*drift with dir_vec=(cos,0,-sin), right_vec=(sin,0,cos)
                nx = (ax + c * fd) + s * fwd_drift;
                nz = (az + (-s) * fd) + c * fwd_drift;
                s_nx[lane] = nx; s_nz[lane] = nz; s_move[lane] = 1;
            }
        }
    }
    __syncthreads();
    if (live && s_move[lane]) {
        // segments are stored transposed (segment-major, env-minor) so that the lanes of a wave - one
        // env each - read consecutive addresses.  No early exit (keeps the loads independent); a
        // segment whose bounding box grown by the radius (+1e-9 guard) excludes the point cannot be
        // within the radius, so skipping it is exact.
        const double px = s_nx[lane], pz = s_nz[lane];
        const double *sg = d.segs + e;
        const size_t N = (size_t)d.N;
        const int ns = d.n_segs[e];
        bool hit = false;
        const double guard = arad + 1e-9;
        for (int i0 = part * 8; i0 < ns; i0 += 8 * STEP_PARTS) {   // 8 segments (32 coalesced loads) in flight, then the tests
            double q[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = (i0 + k < ns) ? i0 + k : ns - 1;   // clamp: re-testing a segment is harmless
#pragma unroll
                for (int c = 0; c < 4; c++) q[k][c] = sg[(size_t)(i * 4 + c) * N];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                bool nearby = px >= fmin(q[k][0], q[k][2]) - guard && px <= fmax(q[k][0], q[k][2]) + guard &&
                              pz >= fmin(q[k][1], q[k][3]) - guard && pz <= fmax(q[k][1], q[k][3]) + guard;
                if (nearby) hit = seg_hit(q[k], px, pz, arad) || hit;
            }
        }
        if (hit) s_hit[lane] = 1;   // every writer stores the same value
    }
    __syncthreads();
    if (part != 0 || !live) return;
    // ---- the rest of MiniWorldEnv.step for this env, sequential (wave 0, lane = env) -------------------------------
    const double max_fwd = d.params[MWB_P_FORWARD_STEP].hi[0];   // self.max_forward_step, miniworld.py:569
    // register-resident lookups by a run-time box index (select chains: no indexed scratch arrays)
    auto pick = [&](const double (&v)[NBX], int i) {
        double r_ = v[0];
#pragma unroll
        for (int b = 1; b < NBX; b++) r_ = (i == b) ? v[b] : r_;
        return r_;
    };
    // MiniWorldEnv.intersect(ent, pos, radius), miniworld.py:933-959, for an entity other than the moving agent's own
    // wall test: walls (all segments, serial: only the carried-entity and pickup tests come here), then the entities
    // in list order except `self_idx` (a box index, or NB for the agent).  Returns 0 none, 1 wall, 2 + k entity k.
    auto intersect_serial = [&](int self_idx, double px, double pz, double radius, bool test_walls) {
        if (test_walls) {
            const int ns = d.n_segs[e];
            for (int i = 0; i < ns; i++) {
                double q[4];
#pragma unroll
                for (int c = 0; c < 4; c++) q[c] = d.segs[(size_t)(i * 4 + c) * d.N + e];
                if (seg_hit(q, px, pz, radius)) return 1;
            }
        }
        int res = 0;
#pragma unroll
        for (int b = NBX - 1; b >= 0; b--) {   // descending, so that the lowest index wins
            const double ddx = bx[b] - px, ddz = bz[b] - pz;
            const bool h = b < NB && b != self_idx && sqrt(ddx * ddx + 0.0 + ddz * ddz) < radius + brad[b];
            res = h ? 2 + b : res;
        }
        if (!res && self_idx != NB) {   // the agent is the last entity of the list
            const double ddx = ax - px, ddz = az - pz;
            if (sqrt(ddx * ddx + 0.0 + ddz * ddz) < radius + arad) res = 2 + NB;
        }
        return res;
    };
    // MiniWorldEnv._get_carry_pos(agent_pos, ent), miniworld.py:594-606, for the carried box
    auto carry_pos = [&](double agx, double agz, double dir, double &cx, double &cy, double &cz) {
        const double er = pick(brad, carried);
        const double dist = (arad + er) + max_fwd;
        const double c = ref_cos(dir), s_ = ref_sin(dir);
        cx = agx + (c * 1.05) * dist; cz = agz + ((-s_) * 1.05) * dist;
        const double eh = pick(bsz, carried);
        double yp = (d.cam[e * 4 + 0] - eh) - 0.3;   // max(self.agent.cam_height - ent.height - 0.3, 0)
        yp = yp > 0 ? yp : 0;
        cy = (0.0 + (0.0 * 1.05) * dist) + 1.0 * yp;
        cx = cx + 0.0 * yp; cz = cz + 0.0 * yp;   // pos + Y_VEC * y_pos
    };
    auto store_carried = [&](double cx, double cy, double cz, bool set_dir) {
#pragma unroll
        for (int b = 0; b < NBX; b++)
            if (b == carried) {
                bx[b] = cx; by[b] = cy; bz[b] = cz;
                const size_t be = (size_t)b * d.N + e;
                d.box_x[be] = cx; d.box_y[be] = cy; d.box_z[be] = cz;
                if (set_dir) d.box_dir[be] = adir;
            }
    };
    if (a == 2 || a == 3) {
        // intersect(self.agent, next_pos, radius): the parallel wall test above, then the boxes in list order
        bool hit = s_hit[lane] != 0 || intersect_serial(NB, nx, nz, arad, false) != 0;
        if (!hit && carried >= 0) {   // the carried entity must fit where it would go (miniworld.py:622-629)
            double cx, cy, cz;
            carry_pos(nx, nz, adir, cx, cy, cz);
            if (intersect_serial(carried, cx, cz, pick(brad, carried), true)) hit = true;
            else store_carried(cx, cy, cz, false);
        }
        if (!hit) { ax = nx; az = nz; d.agent_x[e] = ax; d.agent_z[e] = az; }
    } else if (a == 0 || a == 1) {   // turn_agent, miniworld.py:635-656
        double ta = (a == 0) ? turn_step : -turn_step;
        ta *= (3.141592653589793 / 180);
        const double orig = adir;
        adir += ta;
        if (carried >= 0) {
            double cx, cy, cz;
            carry_pos(ax, az, adir, cx, cy, cz);
            if (intersect_serial(carried, cx, cz, pick(brad, carried), true)) adir = orig;
            else store_carried(cx, cy, cz, true);
        }
        d.agent_dir[e] = adir;
    } else if (a == 4) {   // pickup, miniworld.py:682-689: the first entity within 1.2 r of a point 1.5 r ahead of the agent
        const double c = ref_cos(adir), s_ = ref_sin(adir);
        const double tx = ax + (c * 1.5) * arad, tz = az + ((-s_) * 1.5) * arad;
        const int hit = intersect_serial(NB, tx, tz, 1.2 * arad, true);
        if (carried < 0 && hit >= 2) carried = hit - 2;   // a Box is not static (entity.py:40-46)
    } else if (a == 5) {   // drop, miniworld.py:692-695
        if (carried >= 0) {
#pragma unroll
            for (int b = 0; b < NBX; b++)
                if (b == carried) { by[b] = 0.0; d.box_y[(size_t)b * d.N + e] = 0.0; }
            carried = -1;
        }
    }
    if (carried >= 0) {   // miniworld.py:698-701: the carried entity follows the agent
        double cx, cy, cz;
        carry_pos(ax, az, adir, cx, cy, cz);
        store_carried(cx, cy, cz, true);
    }
    d.carrying[e] = carried;
    double r = 0.0;
    int done = 0;
    if (sc >= d.max_episode_steps) { done = 1; r = 0.0; }   // miniworld.py:708-711
    {   // near(box), miniworld.py:961-971, then the task rule (e.g. envs/maze.py:106-113)
        const double max_forward_step = max_fwd;
        bool near[NBX];
#pragma unroll
        for (int b = 0; b < NBX; b++) {   // np.linalg.norm(ent0.pos - ent1.pos): a carried box is off the floor
            double ddx = bx[b] - ax, ddy = by[b] - 0.0, ddz = bz[b] - az;
            double dist = sqrt((ddx * ddx + ddy * ddy) + ddz * ddz);
            near[b] = b < NB && dist < brad[b] + arad + 1.1 * max_forward_step;
        }
        const double rw = 1.0 - 0.2 * ((double)sc / d.max_episode_steps);   // _reward, miniworld.py:1012
        constexpr int I1 = NBX > 1 ? 1 : 0, I4 = NBX > 5 ? 4 : 0, I5 = NBX > 5 ? 5 : 0;   // box indices of the 2- / 6-box rules
        if (NBX == 2 && d.task == MWB_TASK_TMAZE_TWOBOX) {   // tmaze.py:196-208 / 299-320: goal box first, then the penalty box
            const int g = d.goal_idx[e];
            const bool near_goal = g ? near[I1] : near[0], near_penalty = g ? near[0] : near[I1];
            if (near_goal) { r += rw; done = 1; }
            if (near_penalty) { r += -1 * rw; done = 1; }
            if (d.task_args[0] != 0) {   // the *Features* classes: feature = [near(blue), near(red)], and their step counter
                d.feature[e * 2] = near[I1] ? 1.0f : 0.0f; d.feature[e * 2 + 1] = near[0] ? 1.0f : 0.0f;
                d.task_step_count[e] += 1;
            }
            d.goal_pos[e * 3] = g ? bx[I1] : bx[0]; d.goal_pos[e * 3 + 1] = 0.0; d.goal_pos[e * 3 + 2] = g ? bz[I1] : bz[0];   // info['goal_pos']
        } else if (NBX == 2 && d.task == MWB_TASK_SIM2REAL_PUSH) {   // simtorealpush.py:129-133: the boxes are close enough
            const double ddx = bx[0] - bx[I1], ddz = bz[0] - bz[I1];
            if (sqrt((ddx * ddx + 0.0) + ddz * ddz) < d.goal_dist[e]) { r = 1.0; done = 1; }
        } else if (NBX == 6 && d.task == MWB_TASK_PUTNEXT) {   // putnext.py:45-53: red (box 4) next to yellow (box 5), nothing carried
            if (carried < 0) {
                const double ddx = bx[I4] - bx[I5], ddy = by[I4] - by[I5], ddz = bz[I4] - bz[I5];
                if (sqrt((ddx * ddx + ddy * ddy) + ddz * ddz) < brad[I4] + brad[I5] + 1.1 * max_forward_step) { r += rw; done = 1; }
            }
        } else {
            if (near[0]) { r += rw; done = 1; }
            if (d.task == MWB_TASK_TMAZE || d.task == MWB_TASK_YMAZE) { d.goal_pos[e * 3] = bx[0]; d.goal_pos[e * 3 + 1] = 0.0; d.goal_pos[e * 3 + 2] = bz[0]; }
        }
    }
    d.reward64[e] = r; d.reward[e] = (float)r; d.done[e] = (uint8_t)done; d.ep_steps[e] = sc;
    const bool regen = done && d.auto_reset;   // worker auto-reset, vec_env/subproc_vec_env.py:10-13
    d.reset_set[e] = (uint8_t)regen;
    if (regen) d.reset_list[atomicAdd(d.reset_count, 1)] = e;
}

// ---- MiniWorldEnv.step for the tasks with a general entity list (PickupObjs, RoomObjs, CollectHealth, ThreeRooms, Sign, Sidewalk,
// WallGap): one lane per env, everything serial - these worlds have a handful of wall segments and at most 20 entities, and the
// step is ~2 % of a pass.  Entities live in the [slot][N] arrays (coalesced across lanes); the entity LIST - which PickupObjs
// shortens and CollectHealth re-orders - is the env's row of d.ent_order.  Sums with a mesh's float32 radius follow NumPy >= 2
// (tagged_add).  What a task rule changes after the reference has rendered the step's frame (pickupobjs.py:56-69,
// collecthealth.py:51-64) is recorded as a render override for this pass (d.ovr_slot / d.ovr_pose).
__global__ void __launch_bounds__(64) step_ents_kernel(MwbDev d, const int32_t *__restrict__ actions, const uint8_t *__restrict__ skip) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.order_state[1]) { d.order_state[0] ^= 1; d.order_state[1] = 0; }
    if (e >= d.N) return;
    const size_t N = (size_t)d.N;
    const int E = d.n_boxes;
    if (skip && skip[e]) {   // the fork's 'dummy' command, vec_env/subproc_vec_env.py:26-31
        d.reward[e] = -99.0f; d.reward64[e] = -99.0; d.done[e] = 0; d.ep_steps[e] = d.step_count[e];
        d.reset_set[e] = 0; d.ovr_slot[e] = -1;
        d.feature[e * 2] = 0.0f; d.feature[e * 2 + 1] = 0.0f;
        return;
    }
    int a = actions[(size_t)e * d.act_stride];
    if (d.act_stride == 2 && actions[(size_t)e * 2 + 1] != (a >> 31)) a = -1;   // an int64 outside int32 is no action
    double ax = d.agent_x[e], az = d.agent_z[e], adir = d.agent_dir[e];
    const double arad = d.agent_radius;
    uint8_t *ord = d.ent_order + (size_t)e * MWB_ORDER_STRIDE;
    int n_ord = d.n_order[e];
    int carried = d.carrying[e];
    const double max_fwd = d.params[MWB_P_FORWARD_STEP].hi[0];   // self.max_forward_step
    const double cam_height = d.cam[e * 4 + 0];
    const int ns = d.n_segs[e];
    d.ovr_slot[e] = -1;
    MtSerial g;
    const bool use_rng = d.domain_rand || d.task == MWB_TASK_COLLECTHEALTH;
    if (use_rng) g.load(d.rng + (size_t)e * MWB_MT_WORDS);
    const int sc = d.step_count[e] + 1;   // miniworld.py:663
    d.step_count[e] = sc;
    double fwd_step = d.params[MWB_P_FORWARD_STEP].def[0], fwd_drift = d.params[MWB_P_FORWARD_DRIFT].def[0], turn_step = d.params[MWB_P_TURN_STEP].def[0];
    if (d.domain_rand) {
        fwd_step = g.uniform(d.params[MWB_P_FORWARD_STEP].lo[0], d.params[MWB_P_FORWARD_STEP].hi[0]);
        fwd_drift = g.uniform(d.params[MWB_P_FORWARD_DRIFT].lo[0], d.params[MWB_P_FORWARD_DRIFT].hi[0]);
        turn_step = g.uniform(d.params[MWB_P_TURN_STEP].lo[0], d.params[MWB_P_TURN_STEP].hi[0]);
    }
    auto walls_hit = [&](double px, double pz, double radius) {
        for (int i = 0; i < ns; i++) {
            double q[4];
#pragma unroll
            for (int c = 0; c < 4; c++) q[c] = d.segs[(size_t)(i * 4 + c) * N + e];
            if (seg_hit(q, px, pz, radius)) return true;
        }
        return false;
    };
    // MiniWorldEnv.intersect(ent, pos, radius), miniworld.py:933-959: 0 none, 1 wall, 2 + slot entity, 2 + E the agent
    auto intersect = [&](int self_slot, double px, double pz, double radius, bool rf32) {
        if (walls_hit(px, pz, radius)) return 1;
        for (int k = 0; k < n_ord; k++) {
            const int slot = ord[k];
            if (slot == self_slot) continue;
            double ox, oz, orad; bool of32 = false;
            if (slot == MWB_ENT_AGENT) { ox = ax; oz = az; orad = arad; }
            else { const size_t be = (size_t)slot * N + e; ox = d.box_x[be]; oz = d.box_z[be]; orad = d.ent_radius[be]; of32 = MWB_META_RADF32(d.ent_meta[be]) != 0; }
            const double ddx = ox - px, ddz = oz - pz;
            if (sqrt(ddx * ddx + 0.0 + ddz * ddz) < tagged_add(radius, rf32, orad, of32)) return 2 + (slot == MWB_ENT_AGENT ? E : slot);
        }
        return 0;
    };
    auto carry_pos = [&](int slot, double agx, double agz, double dir, double &cx, double &cy, double &cz) {   // _get_carry_pos, miniworld.py:594-606
        const size_t be = (size_t)slot * N + e;
        bool f32;
        double dist = tagged_add(arad, false, d.ent_radius[be], MWB_META_RADF32(d.ent_meta[be]) != 0, f32);
        dist = tagged_add(dist, f32, max_fwd, false);
        const double c = ref_cos(dir), s_ = ref_sin(dir);
        cx = agx + (c * 1.05) * dist; cz = agz + ((-s_) * 1.05) * dist;
        double yp = (cam_height - d.ent_height[be]) - 0.3;
        yp = yp > 0 ? yp : 0;
        cy = (0.0 + (0.0 * 1.05) * dist) + 1.0 * yp;
        cx = cx + 0.0 * yp; cz = cz + 0.0 * yp;
    };
    auto ent_rad = [&](int slot, bool &f32) { const size_t be = (size_t)slot * N + e; f32 = MWB_META_RADF32(d.ent_meta[be]) != 0; return d.ent_radius[be]; };
    auto set_pose = [&](int slot, double cx, double cy, double cz, bool set_dir) {
        const size_t be = (size_t)slot * N + e;
        d.box_x[be] = cx; d.box_y[be] = cy; d.box_z[be] = cz;
        if (set_dir) d.box_dir[be] = adir;
    };
    if (a == 2 || a == 3) {   // move_agent, miniworld.py:608-633
        const double fd = (a == 2) ? fwd_step : -fwd_step;
        const double c = ref_cos(adir), s_ = ref_sin(adir);
        const double nx = (ax + c * fd) + s_ * fwd_drift, nz = (az + (-s_) * fd) + c * fwd_drift;
        bool hit = intersect(MWB_ENT_AGENT, nx, nz, arad, false) != 0;
        if (!hit && carried >= 0) {
            double cx, cy, cz; bool f32;
            carry_pos(carried, nx, nz, adir, cx, cy, cz);
            const double cr = ent_rad(carried, f32);
            if (intersect(carried, cx, cz, cr, f32)) hit = true;
            else set_pose(carried, cx, cy, cz, false);
        }
        if (!hit) { ax = nx; az = nz; d.agent_x[e] = ax; d.agent_z[e] = az; }
    } else if (a == 0 || a == 1) {   // turn_agent, miniworld.py:635-656
        double ta = (a == 0) ? turn_step : -turn_step;
        ta *= (3.141592653589793 / 180);
        const double orig = adir;
        adir += ta;
        if (carried >= 0) {
            double cx, cy, cz; bool f32;
            carry_pos(carried, ax, az, adir, cx, cy, cz);
            const double cr = ent_rad(carried, f32);
            if (intersect(carried, cx, cz, cr, f32)) adir = orig;
            else set_pose(carried, cx, cy, cz, true);
        }
        d.agent_dir[e] = adir;
    } else if (a == 4) {   // pickup, miniworld.py:682-689
        const double c = ref_cos(adir), s_ = ref_sin(adir);
        const double tx = ax + (c * 1.5) * arad, tz = az + ((-s_) * 1.5) * arad;
        const int hit = intersect(MWB_ENT_AGENT, tx, tz, 1.2 * arad, false);
        if (carried < 0 && hit >= 2 && hit - 2 < E && !MWB_META_STATIC(d.ent_meta[(size_t)(hit - 2) * N + e])) carried = hit - 2;
    } else if (a == 5) {   // drop, miniworld.py:692-695
        if (carried >= 0) { d.box_y[(size_t)carried * N + e] = 0.0; carried = -1; }
    }
    if (carried >= 0) {   // miniworld.py:698-701
        double cx, cy, cz;
        carry_pos(carried, ax, az, adir, cx, cy, cz);
        set_pose(carried, cx, cy, cz, true);
    }
    // ---- obs = self.render_obs() happens here in the reference; the task rules follow
    double r = 0.0;
    int done = 0;
    if (sc >= d.max_episode_steps) { done = 1; r = 0.0; }   // miniworld.py:708-711
    auto near = [&](int slot) {   // MiniWorldEnv.near(ent), miniworld.py:961-971
        const size_t be = (size_t)slot * N + e;
        const double ddx = d.box_x[be] - ax, ddy = d.box_y[be] - 0.0, ddz = d.box_z[be] - az;
        const double dist = sqrt((ddx * ddx + ddy * ddy) + ddz * ddz);
        bool f32;
        double thr = tagged_add(d.ent_radius[be], MWB_META_RADF32(d.ent_meta[be]) != 0, arad, false, f32);
        thr = tagged_add(thr, f32, 1.1 * max_fwd, false);
        return dist < thr;
    };
    auto remove_from_order = [&](int slot) {   // self.entities.remove(ent)
        int k = 0;
        while (k < n_ord && ord[k] != slot) k++;
        for (; k + 1 < n_ord; k++) ord[k] = ord[k + 1];
        n_ord--;
    };
    auto frame_keeps = [&](int slot) {   // the step's frame saw the entity where it is now
        const size_t be = (size_t)slot * N + e;
        d.ovr_slot[e] = slot;
        d.ovr_pose[e * 4 + 0] = d.box_x[be]; d.ovr_pose[e * 4 + 1] = d.box_y[be]; d.ovr_pose[e * 4 + 2] = d.box_z[be]; d.ovr_pose[e * 4 + 3] = d.box_dir[be];
    };
    const double rw = 1.0 - 0.2 * ((double)sc / d.max_episode_steps);   // _reward, miniworld.py:1012
    float feat0 = 0.0f;
    if (d.task == MWB_TASK_PICKUPOBJS) {   // pickupobjs.py:56-69
        if (carried >= 0) {
            frame_keeps(carried);
            remove_from_order(carried);
            d.ent_meta[(size_t)carried * N + e] &= ~(1 << 9);   // no longer in the list
            carried = -1;
            const int np_ = d.task_i[e] + 1;
            d.task_i[e] = np_;
            r = 1.0;
            if (np_ == E) done = 1;
        }
    } else if (d.task == MWB_TASK_COLLECTHEALTH) {   // collecthealth.py:51-77
        double health = d.task_f[e] - 2;
        if (a == 4 && carried >= 0) {   // respawn the kit: entities.remove(kit); place_entity(kit) appends it to the END of the list
            frame_keeps(carried);
            remove_from_order(carried);
            const size_t be = (size_t)carried * N + e;
            const double kr = d.ent_radius[be];
            const bool kf32 = MWB_META_RADF32(d.ent_meta[be]) != 0;
            const double size = d.task_args[0];
            for (int attempt = 0; attempt < 100000; attempt++) {   // place_entity, miniworld.py:876-903, one room
                (void)g.next_double();                                 // choice(rooms, p=room_probs): one double even for a single room
                const double px = g.uniform(0.0 + kr, size - kr);
                (void)g.uniform(0.0, 0.0);
                const double pz = g.uniform(0.0 + kr, size - kr);
                if (!(px > 0.0 && px < size && pz > 0.0 && pz < size)) continue;   // Room.point_inside of the square room
                if (intersect(carried, px, pz, kr, kf32)) continue;
                d.box_x[be] = px; d.box_y[be] = 0.0; d.box_z[be] = pz;
                d.box_dir[be] = g.uniform(-3.141592653589793, 3.141592653589793);
                break;
            }
            ord[n_ord++] = (uint8_t)carried;
            carried = -1;
            health = 100;
        }
        d.task_f[e] = health;
        if (health > 0) r = 2.0;
        else { r = -100.0; done = 1; }
        feat0 = (float)health;   // info['health']
    } else if (d.task == MWB_TASK_SIGN) {   // sign.py:115-128
        if (a == 3) done = 1;   // move_forward + 1: "custom end episode action" (the base step has moved the agent back)
        for (int obj = 0; obj < 2; obj++)
            for (int ci = 0; ci < 3; ci++)
                if (near(obj * 3 + ci)) { done = 1; r = (double)(ci == (int)d.task_args[1] && obj == (int)d.task_args[2]) * 2 - 1; }
    } else if (d.task == MWB_TASK_SIDEWALK) {   // sidewalk.py:74-87: street = the room (0, 6) x (-80, 80)
        if (ax > 0.0 && ax < 6.0 && az > -80.0 && az < 80.0) { r = 0.0; done = 1; }
        if (near(6)) { r += rw; done = 1; }
    } else if (d.task == MWB_TASK_WALLGAP) {   // wallgap.py:54-61
        if (near(0)) { r += rw; done = 1; }
    }   // RoomObjs, ThreeRooms: no rule
    if (use_rng) g.store();
    d.carrying[e] = carried;
    d.n_order[e] = n_ord;
    d.feature[e * 2] = feat0; d.feature[e * 2 + 1] = 0.0f;
    d.reward64[e] = r; d.reward[e] = (float)r; d.done[e] = (uint8_t)done; d.ep_steps[e] = sc;
    const bool regen = done && d.auto_reset;
    d.reset_set[e] = (uint8_t)regen;
    if (regen) d.reset_list[atomicAdd(d.reset_count, 1)] = e;
}

__global__ void mark_reset_kernel(MwbDev d, const uint8_t *__restrict__ mask) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.N) return;
    const bool regen = mask ? (mask[e] != 0) : true;
    d.reset_set[e] = (uint8_t)regen;
    if (regen) d.reset_list[atomicAdd(d.reset_count, 1)] = e;
}

// gen_rot_matrix (math.py:9-23) and numpy's row-vector x matrix product as the reference's np.dot evaluates it
__device__ __forceinline__ void rot_matrix(double axx, double axy, double axz, double angle, double *m) {
    double n = sqrt(axx * axx + axy * axy + axz * axz);
    axx /= n; axy /= n; axz /= n;
    double a = ref_cos(angle / 2.0), s = ref_sin(angle / 2.0);
    double b = -axx * s, c = -axy * s, dd = -axz * s;
    m[0] = a * a + b * b - c * c - dd * dd; m[1] = 2 * (b * c - a * dd); m[2] = 2 * (b * dd + a * c);
    m[3] = 2 * (b * c + a * dd); m[4] = a * a + c * c - b * b - dd * dd; m[5] = 2 * (c * dd - a * b);
    m[6] = 2 * (b * dd - a * c); m[7] = 2 * (c * dd + a * b); m[8] = a * a + dd * dd - b * b - c * c;
}
__device__ __forceinline__ void vec_mat(const double *v, const double *m, double *o) {
    for (int j = 0; j < 3; j++) o[j] = (v[0] * m[j] + v[1] * m[3 + j]) + v[2] * m[6 + j];
}

// ================================================================================= reset kernel
__device__ __forceinline__ void prep_env(const MwbDev &d, int e);   // the prep kernel's per-env body, below
// World generation.  One wave per environment; all 64 lanes run the same sequential logic on
// identical values (the RNG stream is inherently serial) and split the wide parts: the MT19937
// twist, the circle-vs-segments tests of the placement loop and the emission of the room / segment
// tables.  The float64 room store lives in LDS.
struct alignas(16) WRoom {
    double ox[4], oz[4];            // outline, reference order (miniworld.py:732-741, 826)
    double height;
    double p_start[4], p_end[4], p_maxy[4];   // at most one portal per edge in the four tasks
    double min_x, max_x, min_z, max_z, area;
    int n_port[4], nbr[4];
    int tex_fam[3], tex_id[3];      // wall, floor, ceil
    int ne, pad_;                   // num_walls: 4, or 3 for YMaze's triangular hub
};

typedef LDS_AS WRoom LRoom;

enum { TEXF_FLOOR_TILES_BW = 0, TEXF_CONCRETE, TEXF_CONCRETE_TILES, TEXF_BRICK_WALL,
       TEXF_CARDBOARD, TEXF_WOOD, TEXF_WOOD_PLANKS, TEXF_DRYWALL, TEXF_STUCCO, TEXF_CEILING_TILES,
       TEXF_ASPHALT, TEXF_SLIME, TEXF_CINDER_BLOCKS, TEXF_LOGO_MILA };
__constant__ int c_texf_first[14] = {0, 1, 5, 6, 7, 11, 13, 14, 15, 16, 17, 18, 19, 20};
__constant__ int c_texf_count[14] = {1, 4, 1, 1, 4, 2, 1, 1, 1, 1, 1, 1, 1, 1};   // <name>_<i>.png variants, opengl.py:50-58


// |(x, z)| as numpy's norm of (x, 0, z) gives it, sqrt((x x + 0) + z z); an axis-aligned vector needs no arithmetic: sqrt(fl(x x)) == |x|
__device__ __forceinline__ double norm_xz(double x, double z) {
    if (z == 0.0) return fabs(x);
    if (x == 0.0) return fabs(z);
    return sqrt((x * x + 0.0) + z * z);
}
// q / d where d is very often exactly +-1 (an axis-aligned edge direction): q / +-1 == q * +-1
__device__ __forceinline__ double div_unit(double q, double d) { return fabs(d) == 1.0 ? q * d : q / d; }

struct WorldGen {
    LRoom *rooms;
    int n_rooms;
    lds_f64 *cdf;
    double *segs;      // this block's staging rows in HBM (d.seg_stage), S_max x 4: read lane-parallel by walls_hit, transposed into d.segs at the end
    int n_segs;
    lds_i32 *seg_off;
    MtWave rng;
    int lane;
    bool fail;
    // entity tasks: the entities placed so far (slot order = list order while the world is built), in LDS
    lds_f64 *ent_x, *ent_z, *ent_r;
    lds_i32 *ent_f32;
    int n_placed;
    // the one wall that carries a SECOND portal (ThreeRooms: the big room's wall towards both small rooms).  WRoom keeps the
    // opening with the lower start_pos - Room.add_portal sorts the list, miniworld.py:214-215 - this record the other one.
    int xp_room = -1, xp_edge = -1, xp_nbr = -1;
    double xp_start = 0, xp_end = 0, xp_maxy = 0;

    // Room.__init__, miniworld.py:75-138
    __device__ __forceinline__ int add_room(const double *ox, const double *oz, double height, int wall_fam, int floor_fam, int ceil_fam, int ne = 4) {
        LRoom &r = rooms[n_rooms];
        double mnx = ox[0], mxx = ox[0], mnz = oz[0], mxz = oz[0];
        r.ne = ne;
        for (int i = 0; i < 4; i++) {
            const int q = i < ne ? i : 0;   // a triangle's unused 4th slot repeats corner 0 (never read as an edge)
            r.ox[i] = ox[q]; r.oz[i] = oz[q];
            mnx = fmin(mnx, ox[q]); mxx = fmax(mxx, ox[q]); mnz = fmin(mnz, oz[q]); mxz = fmax(mxz, oz[q]);
            r.n_port[i] = 0; r.nbr[i] = -1;
        }
        r.min_x = mnx; r.max_x = mxx; r.min_z = mnz; r.max_z = mxz;
        r.area = (mxx - mnx) * (mxz - mnz);
        r.height = height;
        r.tex_fam[0] = wall_fam; r.tex_fam[1] = floor_fam; r.tex_fam[2] = ceil_fam;
        return n_rooms++;
    }
    // add_rect_room, miniworld.py:718-743
    __device__ __forceinline__ int add_rect_room(double min_x, double max_x, double min_z, double max_z, int wall_fam) {
        double ox[4] = {max_x, max_x, min_x, min_x}, oz[4] = {max_z, min_z, min_z, max_z};
        return add_room(ox, oz, 2.74, wall_fam, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES);
    }
    __device__ __forceinline__ int add_rect_room_ex(double min_x, double max_x, double min_z, double max_z, int wall_fam, int floor_fam, int ceil_fam) {
        double ox[4] = {max_x, max_x, min_x, min_x}, oz[4] = {max_z, min_z, min_z, max_z};
        return add_room(ox, oz, 2.74, wall_fam, floor_fam, ceil_fam);
    }
    // edge direction as Room.__init__ / add_portal compute it: (p1 - p0) / norm
    // Not kept in the room record (96 B x 127 rooms of Maze: the difference between two and three workgroups per CU).  An axis-
    // aligned edge needs no arithmetic: sqrt(fl(x * x)) == |x| and x / |x| == +-1 exactly, 0 / len keeps the zero's sign.
    __device__ __forceinline__ void edge(const LRoom &r, int e, double &dx, double &dz, double &len) const {
        if (e >= r.ne) { dx = 1; dz = 0; len = 1; return; }   // a triangle's unused 4th slot (never read as an edge)
        const int e1 = e + 1 < r.ne ? e + 1 : 0;
        const double ex = r.ox[e1] - r.ox[e], ez = r.oz[e1] - r.oz[e];
        if (ez == 0.0 && ex != 0.0) { len = fabs(ex); dx = copysign(1.0, ex); dz = ez; return; }
        if (ex == 0.0 && ez != 0.0) { len = fabs(ez); dz = copysign(1.0, ez); dx = ex; return; }
        len = sqrt((ex * ex + 0.0) + ez * ez);
        dx = ex / len; dz = ez / len;
    }
    // edge_norms = -cross(edge_dir, Y), normalised (miniworld.py:119-120): (dz, 0, -dx) / its norm - exactly (dz, -dx) for
    // the axis-aligned edges of the rectangle tasks, one more rounding for YMaze's rotated arms
    __device__ __forceinline__ void edge_normal(const LRoom &r, int e, double &nx, double &nz) const {
        double edx_, edz_, elen_;
        edge(r, e, edx_, edz_, elen_);
        const double ex = edz_, ez = -edx_;
        if ((ex == 0.0 && fabs(ez) == 1.0) || (ez == 0.0 && fabs(ex) == 1.0)) { nx = ex; nz = ez; return; }   // norm 1: x / 1 == x
        const double nn = sqrt((ex * ex + 0.0) + ez * ez);
        nx = ex / nn; nz = ez / nn;
    }
    // Room.add_portal, miniworld.py:140-218; mode 0 start/end, 1 min_x/max_x, 2 min_z/max_z
    // returns the slot the opening went to: 0 the room's own record, 1 the second-portal record
    __device__ __forceinline__ int add_portal(int ri, int e, int mode, double a, double b, bool has_max_y, double max_y_in,
                              double &start, double &end) {
        LRoom &r = rooms[ri];
        double dx, dz, len;
        edge(r, e, dx, dz, len);
        if (mode == 1) {
            double m0 = div_unit(a - r.ox[e], dx), m1 = div_unit(b - r.ox[e], dx);
            if (m1 < m0) { double t = m0; m0 = m1; m1 = t; }
            start = m0; end = m1;
        } else if (mode == 2) {
            double m0 = div_unit(a - r.oz[e], dz), m1 = div_unit(b - r.oz[e], dz);
            if (m1 < m0) { double t = m0; m0 = m1; m1 = t; }
            start = m0; end = m1;
        } else { start = a; end = b; }
        if (!(end > start) || !(start >= 0) || !(end <= len)) fail = true;
        const double my = has_max_y ? max_y_in : r.height;
        if (r.n_port[e] == 0) {
            r.p_start[e] = start; r.p_end[e] = end; r.p_maxy[e] = my;
            r.n_port[e] = 1;
            return 0;
        }
        if (r.n_port[e] != 1 || xp_room >= 0) { fail = true; return 0; }   // one wall with two openings per world at most
        xp_room = ri; xp_edge = e;
        r.n_port[e] = 2;
        if (r.p_start[e] > start) {   // stable sort by start_pos: the new opening goes in front
            xp_start = r.p_start[e]; xp_end = r.p_end[e]; xp_maxy = r.p_maxy[e]; xp_nbr = r.nbr[e];
            r.p_start[e] = start; r.p_end[e] = end; r.p_maxy[e] = my; r.nbr[e] = -1;
            return 0;
        }
        xp_start = start; xp_end = end; xp_maxy = my; xp_nbr = -1;
        return 1;
    }
    __device__ __forceinline__ void set_nbr(int ri, int e, int slot, int nb) {
        if (slot == 0) rooms[ri].nbr[e] = nb; else xp_nbr = nb;
    }
    // connect_rooms, miniworld.py:757-843
    __device__ __forceinline__ void connect_rooms(int ia, int ib, int mode, double lo, double hi, bool has_max_y, double max_y) {
        // find_facing_edges (miniworld.py:771-790): first (i, j) in i-major order whose inward normals face
        // each other and whose lines touch.  The 16 candidates are tested by 16 lanes; the lowest set
        // bit of the ballot is the pair the sequential loops would return.
        int idx_a = -1, idx_b = -1;
        {
            const int ci = (lane >> 2) & 3, cj = lane & 3;
            double nax, naz, nbx, nbz;
            edge_normal(rooms[ia], ci, nax, naz);
            edge_normal(rooms[ib], cj, nbx, nbz);
            double dotn = (nax * nbx + 0.0) + naz * nbz;
            double ddx = rooms[ib].ox[cj] - rooms[ia].ox[ci], ddz = rooms[ib].oz[cj] - rooms[ia].oz[ci];
            double dd = (nax * ddx + 0.0) + naz * ddz;
            bool okc = lane < 16 && ci < rooms[ia].ne && cj < rooms[ib].ne && !(dotn > -0.9) && !(dd > 0.05);
            unsigned long long m = __ballot(okc);
            if (m) { int first = __ffsll((long long)m) - 1; idx_a = first >> 2; idx_b = first & 3; }
        }
        if (idx_a < 0) { fail = true; return; }
        double sa, ea, sb, eb;
        const int slot_a = add_portal(ia, idx_a, mode, lo, hi, has_max_y, max_y, sa, ea);
        const int slot_b = add_portal(ib, idx_b, mode, lo, hi, has_max_y, max_y, sb, eb);
        double adx, adz, bdx, bdz, l;
        edge(rooms[ia], idx_a, adx, adz, l);
        edge(rooms[ib], idx_b, bdx, bdz, l);
        const LRoom &A = rooms[ia], &B = rooms[ib];
        double a_x = A.ox[idx_a] + adx * sa, a_z = A.oz[idx_a] + adz * sa;
        double b_x = A.ox[idx_a] + adx * ea, b_z = A.oz[idx_a] + adz * ea;
        double c_x = B.ox[idx_b] + bdx * sb, c_z = B.oz[idx_b] + bdz * sb;
        double d_x = B.ox[idx_b] + bdx * eb, d_z = B.oz[idx_b] + bdz * eb;
        double adx_ = a_x - d_x, adz_ = a_z - d_z;
        if (norm_xz(adx_, adz_) < 0.001) {   // portals directly connected
            set_nbr(ia, idx_a, slot_a, ib); set_nbr(ib, idx_b, slot_b, ia);
            return;
        }
        double len_a = norm_xz(b_x - a_x, b_z - a_z);
        double len_b = norm_xz(d_x - c_x, d_z - c_z);
        double ox[4] = {c_x, b_x, a_x, d_x}, oz[4] = {c_z, b_z, a_z, d_z};
        double my = has_max_y ? max_y : A.height;
        int wf = A.tex_fam[0], ff = A.tex_fam[1], cf = A.tex_fam[2];
        int ic = add_room(ox, oz, my, wf, ff, cf);
        double s0, e0;
        add_portal(ic, 1, 0, 0.0, len_a, false, 0.0, s0, e0);
        add_portal(ic, 3, 0, 0.0, len_b, false, 0.0, s0, e0);
        set_nbr(ia, idx_a, slot_a, ic); set_nbr(ib, idx_b, slot_b, ic);
        rooms[ic].nbr[1] = ia; rooms[ic].nbr[3] = ib;
    }
    // numpy add.reduce pairwise summation (np.sum at miniworld.py:998): up to 128 terms in eight interleaved partial sums ...
    __device__ __forceinline__ double area_sum_block(int first, int n) const {
        if (n < 8) {
            double res = 0.;
            for (int i = 0; i < n; i++) res += rooms[first + i].area;
            return res;
        }
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = rooms[first + k].area;
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += rooms[first + i + k].area;
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += rooms[first + i].area;
        return res;
    }
    // ... above that the range is halved (left half rounded down to a multiple of 8) and the halves' sums are added, left + right.
    // numpy recurses; here the pending ranges sit on a small explicit stack (a recursive member function would keep the whole
    // generator in scratch memory: its `this` escapes).  Depth 8 covers 128 * 2^8 rooms.
    __device__ __forceinline__ double pairwise_area_sum(int first, int n) const {
        if (n <= 128) return area_sum_block(first, n);
        int sf[9], sn[9], phase[9];
        double left[9];
        int sp = 0;
        sf[0] = first; sn[0] = n; phase[0] = 0;
        double ret = 0.0;
        while (sp >= 0) {
            if (sn[sp] <= 128) {   // a leaf: sum it and hand the value to the ranges waiting above
                ret = area_sum_block(sf[sp], sn[sp]);
                sp--;
                while (sp >= 0) {
                    if (phase[sp] == 1) {   // the left half is done: keep it, descend into the right half
                        int n2 = sn[sp] / 2;
                        n2 -= n2 % 8;
                        left[sp] = ret; phase[sp] = 2;
                        sf[sp + 1] = sf[sp] + n2; sn[sp + 1] = sn[sp] - n2; phase[sp + 1] = 0;
                        sp++;
                        break;
                    }
                    ret = left[sp] + ret;
                    sp--;
                }
            } else {
                int n2 = sn[sp] / 2;
                n2 -= n2 % 8;
                phase[sp] = 1;
                sf[sp + 1] = sf[sp]; sn[sp + 1] = n2; phase[sp + 1] = 0;
                sp++;
            }
        }
        return ret;
    }
    // number of collidable segments edge e of room r produces (gen_seg_poly calls, miniworld.py:312-375)
    __device__ int edge_segs(const LRoom &r, int e, double *out /* may be null */) const {
        double dx, dz, len;
        edge(r, e, dx, dz, len);
        int n = 0;
        double st[3], en[3];
        if (r.n_port[e] == 0) { st[0] = 0; en[0] = len; n = 1; }
        else if (r.n_port[e] == 1) {
            st[0] = 0; en[0] = r.p_start[e];
            st[1] = r.p_end[e]; en[1] = len;
            n = 2;
        } else {   // two openings (this is the room of the second-portal record): wall, opening, wall, opening, wall
            st[0] = 0; en[0] = r.p_start[e];
            st[1] = r.p_end[e]; en[1] = xp_start;
            st[2] = xp_end; en[2] = len;
            n = 3;
        }
        int cnt = 0;
        for (int k = 0; k < n; k++) {
            if (en[k] == st[k]) continue;   // gen_seg_poly: seg_end == seg_start
            if (out) {
                // wall_segs.append([s_p1, s_p0]), s_p = edge_p0 + seg * side_vec (miniworld.py:281-286)
                double p0x = r.ox[e] + st[k] * dx, p0z = r.oz[e] + st[k] * dz;
                double p1x = r.ox[e] + en[k] * dx, p1z = r.oz[e] + en[k] * dz;
                out[cnt * 4 + 0] = p1x; out[cnt * 4 + 1] = p1z; out[cnt * 4 + 2] = p0x; out[cnt * 4 + 3] = p0z;
            }
            cnt++;
        }
        return cnt;
    }
    // MiniWorldEnv._gen_static_data, miniworld.py:981-998 (+ Room._gen_static_data 243-245, 311-375)
    __device__ __forceinline__ void gen_static_data(bool use_rng, int S_max) {
        if (use_rng) {   // Texture.get, opengl.py:40-69: rng.int(0, n_variants) per room and slot, in order
            for (int i = 0; i < n_rooms; i++)
                for (int k = 0; k < 3; k++) {
                    int fam = rooms[i].tex_fam[k];
                    int idx = rng.randint(0, c_texf_count[fam]);
                    __syncthreads();
                    if (lane == 0) rooms[i].tex_id[k] = c_texf_first[fam] + idx;
                }
        } else {
            for (int i = lane; i < n_rooms; i += WAVE)
                for (int k = 0; k < 3; k++) rooms[i].tex_id[k] = c_texf_first[rooms[i].tex_fam[k]];
        }
        __syncthreads();
        for (int i = lane; i < n_rooms; i += WAVE) {   // segments per room, lane-parallel
            int c = 0;
            for (int e = 0; e < rooms[i].ne; e++) c += edge_segs(rooms[i], e, nullptr);
            seg_off[i] = c;
        }
        __syncthreads();
        int off = 0;
        for (int i = 0; i < n_rooms; i++) {   // exclusive prefix in room order (the reference's segment order)
            int c = seg_off[i];
            __syncthreads();
            if (lane == 0) seg_off[i] = off;
            off += c;
        }
        __syncthreads();
        n_segs = off;
        if (n_segs > S_max) { fail = true; n_segs = S_max; }
        __syncthreads();
        for (int i = lane; i < n_rooms && !fail; i += WAVE) {
            int o = seg_off[i];
            for (int e = 0; e < rooms[i].ne; e++) o += edge_segs(rooms[i], e, segs + o * 4);
        }
        __syncthreads();
        double sum = pairwise_area_sum(0, n_rooms);
        // room_probs = area / sum (miniworld.py:997-998); RandomState.choice(p): cdf = p.cumsum(); cdf /= cdf[-1]
        for (int i = lane; i < n_rooms; i += WAVE) cdf[i] = rooms[i].area / sum;   // the divisions, lane-parallel
        __syncthreads();
        double acc = 0;
        for (int i = 0; i < n_rooms; i++) {   // sequential cumsum, every lane the same
            double p = cdf[i];
            acc = (i == 0) ? p : acc + p;
            __syncthreads();
            if (lane == 0) cdf[i] = acc;
        }
        __syncthreads();
        double last = cdf[n_rooms - 1];
        __syncthreads();
        for (int i = lane; i < n_rooms; i += WAVE) cdf[i] = cdf[i] / last;
        __syncthreads();
    }
    // MiniWorldEnv.intersect walls part, all lanes
    __device__ __forceinline__ bool walls_hit(double px, double pz, double radius) const {
        bool h = false;
        for (int i = lane; i < n_segs; i += WAVE) h = h || seg_hit(segs + i * 4, px, pz, radius);
        return __any(h);
    }
    // place_entity, miniworld.py:845-907 (pos=None path).  opt.room < 0: room=None, i.e. drawn with
    // choice(rooms, p=room_probs); others: the entities placed before (list order), all of radius other_radius
    struct PlaceOpt {
        int room = -1;
        bool has_dir = false, has_min_x = false, has_max_x = false, has_min_z = false, has_max_z = false;
        double dir = 0, min_x = 0, max_x = 0, min_z = 0, max_z = 0;
        int n_others = 0;
        double other_x[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0}, other_z[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0}, other_radius = 0;
        double other_radius2 = -1;   // radius of the second other entity when it differs (sim-to-real boxes)
        double other_r[MWB_MAX_BOXES] = {-1, -1, -1, -1, -1, -1};   // per-entity radii where given (>= 0), else the two above
        bool lds_ents = false, self_f32 = false;   // entity tasks: the others are WorldGen::ent_* [0, n_placed); radius is a float32 scalar
    };
    __device__ __forceinline__ void place_entity(double radius, bool has_other, double other_x, double other_z, double other_radius,
                                 bool has_dir, double dir_in, bool has_min_x, double min_x, bool has_max_x, double max_x,
                                 double &out_x, double &out_z, double &out_dir) {
        PlaceOpt o;
        o.has_dir = has_dir; o.dir = dir_in; o.has_min_x = has_min_x; o.min_x = min_x; o.has_max_x = has_max_x; o.max_x = max_x;
        o.n_others = has_other ? 1 : 0; o.other_x[0] = other_x; o.other_z[0] = other_z; o.other_radius = other_radius;
        place_entity_ex(radius, o, out_x, out_z, out_dir);
    }
    __device__ __forceinline__ void place_entity_ex(double radius, const PlaceOpt &opt, double &out_x, double &out_z, double &out_dir) {
        const bool has_dir = opt.has_dir;
        const double dir_in = opt.dir;
        for (int attempt = 0; attempt < 100000; attempt++) {
            int lo = opt.room;
            if (lo < 0) {
                double u = rng.next_double();   // choice(rooms, p=room_probs): searchsorted(cdf, u, 'right')
                int hi = n_rooms;
                lo = 0;
                while (lo < hi) { int mid = (lo + hi) / 2; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
            }
            const LRoom &r = rooms[lo < n_rooms ? lo : n_rooms - 1];
            double lx = opt.has_min_x ? opt.min_x : r.min_x, hx = opt.has_max_x ? opt.max_x : r.max_x;
            double lz = opt.has_min_z ? opt.min_z : r.min_z, hz = opt.has_max_z ? opt.max_z : r.max_z;
            double px = rng.uniform(lx + radius, hx - radius);
            (void)rng.uniform(0.0, 0.0);   // the y component draws too
            double pz = rng.uniform(lz + radius, hz - radius);
            // Room.point_inside, miniworld.py:220-232
            bool inside = true;
            for (int e = 0; e < r.ne; e++) {
                double nx, nz;
                edge_normal(r, e, nx, nz);
                double dot = (nx * (px - r.ox[e]) + 0.0) + nz * (pz - r.oz[e]);
                if (!(dot > 0)) inside = false;
            }
            if (!inside) continue;
            if (walls_hit(px, pz, radius)) continue;
            bool blocked = false;
            for (int k = 0; k < opt.n_others; k++) {
                double ddx = opt.other_x[k] - px, ddz = opt.other_z[k] - pz;
                double dist = sqrt(ddx * ddx + 0.0 + ddz * ddz);
                const double orad = opt.other_r[k] >= 0 ? opt.other_r[k] : ((k == 1 && opt.other_radius2 >= 0) ? opt.other_radius2 : opt.other_radius);
                if (dist < radius + orad) blocked = true;
            }
            if (opt.lds_ents)
                for (int k = 0; k < n_placed; k++) {
                    double ddx = ent_x[k] - px, ddz = ent_z[k] - pz;
                    double dist = sqrt(ddx * ddx + 0.0 + ddz * ddz);
                    if (dist < tagged_add(radius, opt.self_f32, ent_r[k], ent_f32[k] != 0)) blocked = true;
                }
            if (blocked) continue;
            out_dir = has_dir ? dir_in : rng.uniform(-3.141592653589793, 3.141592653589793);
            out_x = px; out_z = pz;
            return;
        }
        fail = true;
    }
};

__device__ __forceinline__ void sample_param(MtWave &g, const MwbParam &p, int n, bool use_rng, double *out) {
    for (int k = 0; k < n; k++) out[k] = use_rng ? g.uniform(p.lo[k], p.hi[k]) : p.def[k];   // params.py:81-99
}

// One instantiation per task: the task is a compile-time constant, so every other task's world generation - and the registers /
// scratch its locals would claim - is gone from the kernel (one kernel with every branch: 256 VGPRs + 256 AGPRs + 2.3 KB/lane of scratch).
template <int TASK_>
__global__ void __launch_bounds__(WAVE) reset_kernel(MwbDev d) {
    // The regenerated envs' chain (this kernel, then their render) runs beside the bulk render, whose five waves per SIMD
    // would otherwise leave a lone latency-bound wave one issue slot in six: raise the wave's priority at the arbiter
    __builtin_amdgcn_s_setprio(3);
    const int count = d.reset_count[0];
    const int lane = threadIdx.x;
    for (int li = blockIdx.x; li < count; li += gridDim.x) {
    const int e = d.reset_list[li];
    __syncthreads();   // LDS is reused from the previous env of this block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LRoom *rooms = (LRoom *)smem;
    size_t off = (size_t)d.R_max * sizeof(WRoom);
    lds_f64 *cdf = (lds_f64 *)(smem + off); off += (size_t)d.R_max * sizeof(double);
    lds_i32 *seg_off = (lds_i32 *)(smem + off); off += (size_t)((d.R_max + 3) & ~3) * sizeof(int);
    lds_u32 *key = (lds_u32 *)(smem + off); off += 624 * sizeof(uint32_t);
    // Maze: the frames of the depth-first search (3 ints per cell + 1) and the visited flags; entity tasks: per slot x y z dir size
    // radius height scale bias[3] (f64) and meta, radius-is-float32, colour index (i32), text textures.  (mwb_reset_lds_bytes)
    lds_i32 *dfs = (lds_i32 *)(smem + off);
    lds_f64 *E_x = (lds_f64 *)(smem + off), *E_y = E_x + MWB_MAX_ENTS, *E_z = E_y + MWB_MAX_ENTS, *E_dir = E_z + MWB_MAX_ENTS,
           *E_size = E_dir + MWB_MAX_ENTS, *E_rad = E_size + MWB_MAX_ENTS, *E_hgt = E_rad + MWB_MAX_ENTS, *E_scale = E_hgt + MWB_MAX_ENTS,
           *E_bias = E_scale + MWB_MAX_ENTS;
    lds_i32 *E_meta = (lds_i32 *)(E_bias + 3 * MWB_MAX_ENTS), *E_f32 = E_meta + MWB_MAX_ENTS, *E_col = E_f32 + MWB_MAX_ENTS, *E_text = E_col + MWB_MAX_ENTS;
    // The collision segments are staged in HBM, one row set per block (8 KB of Maze's LDS: the difference between three and four
    // workgroups per CU); the placement loop reads them lane-parallel, coalesced, out of L1 / L2.
    double *segs = d.seg_stage + (size_t)blockIdx.x * d.S_max * 4;

    uint32_t *st = d.rng + (size_t)e * MWB_MT_WORDS;
    for (int i = lane; i < 624; i += WAVE) key[i] = st[i];
    __syncthreads();

    WorldGen w;
    w.rooms = rooms; w.n_rooms = 0; w.cdf = cdf; w.segs = segs; w.n_segs = 0; w.seg_off = seg_off;
    w.rng.key = key; w.rng.pos = (int)st[624]; w.rng.lane = lane; w.lane = lane; w.fail = false;
    w.ent_x = E_x; w.ent_z = E_z; w.ent_r = E_rad; w.ent_f32 = E_f32; w.n_placed = 0;
    bool dr = d.domain_rand != 0;
    double task_f0 = 0.0;   // CollectHealth.health after the reset

    double box_x = 0, box_z = 0, box_dir = 0, ag_x = 0, ag_z = 0, ag_dir = 0;
    double box2_x = 0, box2_z = 0, box2_dir = 0;
    double brad = box_radius(0.8), brad2 = brad, box_s = 0.8, box2_s = d.n_boxes > 1 ? 0.8 : 0.0, goal_dist = 0.0;
    double pn_x[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0}, pn_z[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0}, pn_dir[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0},
           pn_s[MWB_MAX_BOXES] = {0, 0, 0, 0, 0, 0};   // PutNext's six boxes
    const double arad = d.agent_radius;
    // reset() overrides of the T-maze family run before MiniWorldEnv.reset (every lane computes the same)
    int goal_idx = d.goal_idx[e];
    long long episode_count = d.episode_count[e];
    if (TASK_ == MWB_TASK_TMAZE && d.task_args[3] > 0) {   // TMazeDynamic.reset, tmaze.py:98-105
        episode_count += 1;
        if (episode_count % (long long)d.task_args[3] == 0) goal_idx = (goal_idx + 1) % 2;
    } else if (TASK_ == MWB_TASK_TMAZE_TWOBOX) {
        if (d.task_args[0] == 0) {   // TMazeTwoBoxDynamic.reset, tmaze.py:210-217
            episode_count += 1;
            if (episode_count % (long long)d.task_args[3] == 0) goal_idx = (goal_idx + 1) % 2;
        } else if ((double)d.task_step_count[e] > d.task_args[3]) {   // *Features*.reset, tmaze.py:322-330; the counter is
            goal_idx = (goal_idx + 1) % 2;                               // never cleared (the reference assigns a misspelt name)
        }
    }
    if (TASK_ == MWB_TASK_SIM2REAL_GOTO || TASK_ == MWB_TASK_SIM2REAL_PUSH) {   // envs/simtorealgoto.py:40-82, simtorealpush.py:39-107
        const bool push = TASK_ == MWB_TASK_SIM2REAL_PUSH;
        const double size = push ? w.rng.uniform(1.6, 1.7) : w.rng.uniform(1, 2);
        const double wall_height = push ? w.rng.uniform(0.42, 0.50) : w.rng.uniform(0.20, 0.50);
        box_s = push ? w.rng.uniform(0.075, 0.090) : w.rng.uniform(0.07, 0.12);
        box2_s = push ? w.rng.uniform(0.075, 0.090) : 0.0;
        brad = box_radius(box_s); brad2 = box_radius(box2_s);
        // RandGen.choice -> np_random.choice(len) -> randint (random.py:33-41)
        const int fi = w.rng.randint(0, 3);
        const int floor_fam = fi == 0 ? TEXF_CARDBOARD : fi == 1 ? TEXF_WOOD : TEXF_WOOD_PLANKS;
        int wall_fam;
        if (push) {
            const int wi = w.rng.randint(0, 4);
            wall_fam = wi == 0 ? TEXF_DRYWALL : wi == 1 ? TEXF_STUCCO : wi == 2 ? TEXF_CONCRETE_TILES : TEXF_CEILING_TILES;
        } else {
            const int wi = w.rng.randint(0, 5);
            wall_fam = wi == 0 ? TEXF_DRYWALL : wi == 1 ? TEXF_STUCCO : wi == 2 ? TEXF_CARDBOARD : wi == 3 ? TEXF_CONCRETE_TILES : TEXF_CEILING_TILES;
        }
        double ox[4] = {size, size, 0, 0}, oz[4] = {size, 0, 0, size};
        w.add_room(ox, oz, wall_height, wall_fam, floor_fam, TEXF_CONCRETE_TILES);   // no ceiling: negative height word in the room table
        w.gen_static_data(dr, d.S_max);
        WorldGen::PlaceOpt o;
        if (push) {
            goal_dist = 1.5 * (box_s + box2_s);
            const double min_pos = 2 * d.params[MWB_P_BOT_RADIUS].hi[0], max_pos = size - 2 * d.params[MWB_P_BOT_RADIUS].hi[0];
            o.has_min_x = o.has_max_x = o.has_min_z = o.has_max_z = true;
            o.min_x = o.min_z = min_pos; o.max_x = o.max_z = max_pos;
            for (int attempt = 0; attempt < 100000; attempt++) {   // boxes can't start too close to each other
                o.n_others = 0;
                w.place_entity_ex(brad, o, box_x, box_z, box_dir);
                o.n_others = 1; o.other_x[0] = box_x; o.other_z[0] = box_z; o.other_radius = brad;
                w.place_entity_ex(brad2, o, box2_x, box2_z, box2_dir);
                const double ddx = box_x - box2_x, ddz = box_z - box2_z;
                if (sqrt((ddx * ddx + 0.0) + ddz * ddz) > goal_dist) break;
                if (attempt == 99999) w.fail = true;
            }
        } else {
            w.place_entity_ex(brad, o, box_x, box_z, box_dir);
        }
        WorldGen::PlaceOpt a;
        a.n_others = d.n_boxes; a.other_radius = brad; a.other_radius2 = brad2;
        a.other_x[0] = box_x; a.other_z[0] = box_z; a.other_x[1] = box2_x; a.other_z[1] = box2_z;
        w.place_entity_ex(arad, a, ag_x, ag_z, ag_dir);
    } else if (TASK_ == MWB_TASK_TMAZE || TASK_ == MWB_TASK_TMAZE_TWOBOX) {   // envs/tmaze.py:27-61, 151-194
        w.add_rect_room(-1, 8, -2, 2, TEXF_CONCRETE);
        int r2 = w.add_rect_room(8, 12, -8, 8, TEXF_CONCRETE);
        const double r2min = rooms[r2].min_z, r2max = rooms[r2].max_z;
        w.connect_rooms(0, 1, 2, -2, 2, false, 0);
        // `if self.rand.bool():` (tmaze.py:53) is evaluated before the first place_entity, i.e. before the
        // texture draws of _gen_static_data (miniworld.py:865-866)
        const bool random_arm = TASK_ == MWB_TASK_TMAZE && d.task_args[0] == 0;
        const bool left_arm = random_arm && w.rng.randint(0, 2) == 0;   // RandGen.bool, random.py:26-31
        w.gen_static_data(dr, d.S_max);
        WorldGen::PlaceOpt o;
        if (TASK_ == MWB_TASK_TMAZE_TWOBOX) {   // red box at (10, -6), blue box at (10, 6): min == max
            o.has_min_x = o.has_max_x = o.has_min_z = o.has_max_z = true;
            o.min_x = o.max_x = 10; o.min_z = o.max_z = -6;
            w.place_entity_ex(brad, o, box_x, box_z, box_dir);
            o.min_z = o.max_z = 6;
            o.n_others = 1; o.other_x[0] = box_x; o.other_z[0] = box_z; o.other_radius = brad;
            w.place_entity_ex(brad, o, box2_x, box2_z, box2_dir);
        } else if (d.task_args[0] != 0) {   // goal_pos given (TMazeLeft / Right / Dynamic)
            double gx = d.task_args[1], gz = d.task_args[2];
            if (d.task_args[3] > 0) { gx = 10; gz = goal_idx ? 6 : -6; }   // TMazeDynamic.goals, tmaze.py:88
            o.has_min_x = o.has_max_x = o.has_min_z = o.has_max_z = true;
            o.min_x = o.max_x = gx; o.min_z = o.max_z = gz;
            w.place_entity_ex(brad, o, box_x, box_z, box_dir);
        } else {
            o.room = r2;
            if (left_arm) { o.has_max_z = true; o.max_z = r2min + 2; }
            else { o.has_min_z = true; o.min_z = r2max - 2; }
            w.place_entity_ex(brad, o, box_x, box_z, box_dir);
        }
        WorldGen::PlaceOpt a;
        a.room = 0; a.has_dir = true;
        a.dir = w.rng.uniform(-3.141592653589793 / 4, 3.141592653589793 / 4);   // drawn before the placement loop
        a.n_others = d.n_boxes; a.other_radius = brad;
        a.other_x[0] = box_x; a.other_z[0] = box_z; a.other_x[1] = box2_x; a.other_z[1] = box2_z;
        w.place_entity_ex(arad, a, ag_x, ag_z, ag_dir);
    } else if (TASK_ == MWB_TASK_PUTNEXT) {   // envs/putnext.py:22-43
        const double size = d.task_args[0];
        w.add_rect_room(0, size, 0, size, TEXF_CONCRETE);
        WorldGen::PlaceOpt o;
        for (int b = 0; b < 6; b++) {   // for color in COLOR_NAMES: Box(color, size=self.rand.float(0.6, 0.85)); place_entity(box)
            pn_s[b] = w.rng.uniform(0.6, 0.85);            // drawn before the placement (and, for the first box, before
            if (b == 0) w.gen_static_data(dr, d.S_max);    // the texture draws of _gen_static_data, miniworld.py:865-866)
            const double rb = box_radius(pn_s[b]);
            o.n_others = b;
            w.place_entity_ex(rb, o, pn_x[b], pn_z[b], pn_dir[b]);
            o.other_x[b] = pn_x[b]; o.other_z[b] = pn_z[b]; o.other_r[b] = rb;
        }
        o.n_others = 6;
        w.place_entity_ex(arad, o, ag_x, ag_z, ag_dir);
    } else if (TASK_ == MWB_TASK_YMAZE) {   // envs/ymaze.py:28-83
        const double mox[4] = {-9.15, -9.15, -1.15, -1.15}, moz[4] = {-2, 2, 2, -2};
        w.add_room(mox, moz, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES);
        const double hox[4] = {-1.15, -1.15, 2.31, 0}, hoz[4] = {-2, 2, 0, 0};   // the hub: a triangle
        w.add_room(hox, hoz, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES, 3);
        for (int arm = 0; arm < 2; arm++) {   // np.dot(main_outline, gen_rot_matrix(Y_VEC, -+120 deg))
            double m[9], ox[4], oz[4];
            rot_matrix(0, 1, 0, (arm == 0 ? -120 : 120) * (3.141592653589793 / 180), m);
            for (int i = 0; i < 4; i++) {
                const double v[3] = {mox[i], 0, moz[i]};
                double out[3];
                vec_mat(v, m, out);
                ox[i] = out[0]; oz[i] = out[2];
            }
            w.add_room(ox, oz, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES);
        }
        w.connect_rooms(0, 1, 2, -2, 2, false, 0);
        w.connect_rooms(2, 1, 2, -1.995, 0, false, 0);
        w.connect_rooms(3, 1, 2, 0, 1.995, false, 0);
        const double arm2_min_z = rooms[2].min_z, arm3_max_z = rooms[3].max_z;
        // `if self.rand.bool():` (ymaze.py:69) comes before the first place_entity, i.e. before the texture draws
        const bool random_arm = d.task_args[0] == 0;
        const bool left_arm = random_arm && w.rng.randint(0, 2) == 0;
        w.gen_static_data(dr, d.S_max);
        WorldGen::PlaceOpt o;
        if (!random_arm) {   // goal_pos given (YMazeLeft / YMazeRight, ymaze.py:97-103): min == max, room drawn at random
            o.has_min_x = o.has_max_x = o.has_min_z = o.has_max_z = true;
            o.min_x = o.max_x = d.task_args[1]; o.min_z = o.max_z = d.task_args[2];
        } else if (left_arm) { o.room = 2; o.has_max_z = true; o.max_z = arm2_min_z + 2.5; }
        else { o.room = 3; o.has_min_z = true; o.min_z = arm3_max_z - 2.5; }
        w.place_entity_ex(brad, o, box_x, box_z, box_dir);
        WorldGen::PlaceOpt a;
        a.room = 0; a.has_dir = true;
        a.dir = w.rng.uniform(-3.141592653589793 / 4, 3.141592653589793 / 4);
        a.n_others = 1; a.other_radius = brad; a.other_x[0] = box_x; a.other_z[0] = box_z;
        w.place_entity_ex(arad, a, ag_x, ag_z, ag_dir);
    } else if ((TASK_ >= MWB_TASK_PICKUPOBJS)) {
        // ---- tasks with a general entity list.  Every lane runs the same serial logic; lane 0 records the entities in LDS.
        auto put = [&](int b, int kind, int geom, bool is_static, int color, double size, double radius, bool f32, double height, double scale) {
            __syncthreads();
            if (lane == 0) {
                E_meta[b] = kind | (geom << 4) | ((is_static ? 1 : 0) << 8) | (1 << 9) | ((f32 ? 1 : 0) << 10) | ((color + 1) << 12);
                E_size[b] = size; E_rad[b] = radius; E_f32[b] = f32 ? 1 : 0; E_hgt[b] = height; E_scale[b] = scale; E_col[b] = color; E_y[b] = 0.0;
            }
            __syncthreads();
        };
        auto put_box = [&](int b, double size, int color) { put(b, MWB_ENT_BOX, 0, false, color, size, box_radius(size), false, size, 0.0); };
        auto put_mesh = [&](int b, int geom, double height, bool is_static, int color) {   // MeshEnt.__init__, dimensions from the host's table
            double sc = 0, rad = 0; bool f32 = false, found = false;
            for (int k = 0; k < d.n_mesh_dims; k++)
                if (d.mesh_dims[k].geom == geom && d.mesh_dims[k].height == height) { sc = d.mesh_dims[k].scale; rad = d.mesh_dims[k].radius; f32 = d.mesh_dims[k].is_f32 != 0; found = true; }
            if (!found) w.fail = true;
            put(b, MWB_ENT_MESH, geom, is_static, color, height, rad, f32, height, sc);
        };
        auto pose = [&](int b, double x, double y, double z, double dir) {
            __syncthreads();
            if (lane == 0) { E_x[b] = x; E_y[b] = y; E_z[b] = z; E_dir[b] = dir; }
            __syncthreads();
            w.n_placed = b + 1;
        };
        auto place = [&](int b, WorldGen::PlaceOpt &o) {   // place_entity(ent): among everything placed so far
            o.lds_ents = true; o.self_f32 = E_f32[b] != 0;
            double x, z, dir;
            w.place_entity_ex(E_rad[b], o, x, z, dir);
            pose(b, x, 0.0, z, dir);
        };
        auto place_at = [&](int b, double x, double y, double z, bool has_dir, double dir) {   // place_entity(ent, pos=...), miniworld.py:869-873
            const double dd = has_dir ? dir : w.rng.uniform(-3.141592653589793, 3.141592653589793);
            pose(b, x, y, z, dd);
        };
        auto place_agent = [&](WorldGen::PlaceOpt &o, double radius) {
            o.lds_ents = true; o.self_f32 = false;
            w.place_entity_ex(radius, o, ag_x, ag_z, ag_dir);
        };
        const int E = d.n_boxes;
        if (TASK_ == MWB_TASK_PICKUPOBJS) {   // envs/pickupobjs.py:28-54
            const double size = d.task_args[0];
            w.add_rect_room_ex(0, size, 0, size, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES);
            for (int b = 0; b < E; b++) {
                const int type = w.rng.randint(0, 3);    // self.rand.choice([Ball, Box, Key])
                const int color = w.rng.randint(0, 6);   // self.rand.color()
                if (type == 1) put_box(b, 0.9, color);
                else if (type == 0) put_mesh(b, MWB_MESH_BALL, 0.9, false, color);
                else put_mesh(b, MWB_MESH_KEY, 0.35, false, color);
                if (b == 0) w.gen_static_data(dr, d.S_max);   // the first place_entity (after the constructor's draws)
                WorldGen::PlaceOpt o;
                place(b, o);
            }
            WorldGen::PlaceOpt a;
            place_agent(a, arad);
        } else if (TASK_ == MWB_TASK_ROOMOBJS) {   // envs/roomobjs.py:24-47 (agent.radius = 1.5: d.agent_radius)
            const double size = d.task_args[0];
            w.add_rect_room_ex(0, size, 0, size, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES);
            for (int b = 0; b < 3; b++) {
                const int color = w.rng.randint(0, 6);   // an argument of the entity's constructor: drawn before the placement
                if (b == 0) put_box(b, 0.9, color);
                else if (b == 1) put_mesh(b, MWB_MESH_BALL, 0.9, false, color);
                else put_mesh(b, MWB_MESH_KEY, 0.35, false, color);
                if (b == 0) w.gen_static_data(dr, d.S_max);
                WorldGen::PlaceOpt o;
                place(b, o);
            }
            WorldGen::PlaceOpt a;
            place_agent(a, arad);
        } else if (TASK_ == MWB_TASK_COLLECTHEALTH) {   // envs/collecthealth.py:28-49
            const double size = d.task_args[0];
            w.add_rect_room_ex(0, size, 0, size, TEXF_CINDER_BLOCKS, TEXF_SLIME, TEXF_CONCRETE_TILES);
            w.gen_static_data(dr, d.S_max);
            for (int b = 0; b < 18; b++) {
                put_mesh(b, MWB_MESH_MEDKIT, 0.40, false, -1);
                WorldGen::PlaceOpt o;
                place(b, o);
            }
            WorldGen::PlaceOpt a;
            place_agent(a, arad);
            task_f0 = 100.0;
        } else if (TASK_ == MWB_TASK_THREEROOMS) {   // envs/threerooms.py:22-69
            w.add_rect_room(-7, 7, 0.5, 7, TEXF_CONCRETE);
            w.add_rect_room(-7, -1, -7, -0.5, TEXF_CONCRETE);
            w.add_rect_room(1, 7, -7, -0.5, TEXF_CONCRETE);
            w.connect_rooms(0, 1, 1, -5.25, -2.75, false, 0);
            w.connect_rooms(0, 2, 1, 2.75, 5.25, false, 0);
            w.gen_static_data(dr, d.S_max);
            WorldGen::PlaceOpt o;
            put_box(0, 0.8, 4); place(0, o);
            put_box(1, 0.6, 1); place(1, o);
            {   // ImageFrame(pos=[0, 1.35, 7], dir=pi/2, width=1.8, tex_name='logo_mila') appended: radius 0; size = its height
                const MwbTexDesc &T = d.tex_desc[c_texf_first[TEXF_LOGO_MILA]];
                const double fh = ((double)T.h / T.w) * 1.8;
                put(2, MWB_ENT_IMAGE, 0, true, -1, 1.8, 0.0, false, fh, 0.0);
                pose(2, 0.0, 1.35, 7.0, 3.141592653589793 / 2);
            }
            put_mesh(3, MWB_MESH_DUCKIE, 0.25, false, -1); place(3, o);
            put_mesh(4, MWB_MESH_KEY, 0.35, false, 0); place(4, o);
            put_mesh(5, MWB_MESH_BALL, 0.6, false, 1); place(5, o);
            WorldGen::PlaceOpt a;
            place_agent(a, arad);
        } else if (TASK_ == MWB_TASK_SIGN) {   // envs/sign.py:75-113
            const double size = d.task_args[0], gap = 0.25;
            w.add_rect_room(0, size, 0, size * 0.65, TEXF_CONCRETE);
            w.add_rect_room(0, size * 3 / 5, size * 0.65 + gap, size * 1.3, TEXF_CONCRETE);
            w.add_rect_room(size * 3 / 5, size, size * 0.65 + gap, size * 1.3, TEXF_CONCRETE);
            w.connect_rooms(0, 1, 1, 0, size * 3 / 5, false, 0);
            w.connect_rooms(1, 2, 2, size * 0.65 + gap, size * 1.3, false, 0);
            w.gen_static_data(dr, d.S_max);
            const int box_col[3] = {0, 4, 1};   // blue, red, green
            const double box_at[3][2] = {{1, 1}, {9, 1}, {9, 5}}, key_at[3][2] = {{5, 1}, {1, 5}, {1, 9}};
            for (int i = 0; i < 3; i++) { put_box(i, 0.8, box_col[i]); place_at(i, box_at[i][0], 0.0, box_at[i][1], false, 0); }
            for (int i = 0; i < 3; i++) { put_mesh(3 + i, MWB_MESH_KEY, 0.6, false, box_col[i]); place_at(3 + i, key_at[i][0], 0.0, key_at[i][1], false, 0); }   // BigKey
            {   // TextFrame(pos=[size, 1.35, size + gap], dir=pi, str, height=1): width = len(str) * height
                const int ci = (int)d.task_args[1];
                const int n_ch = ci == 0 ? 4 : ci == 1 ? 3 : 5;   // "BLUE" "RED" "GREEN"
                put(6, MWB_ENT_TEXT, 0, true, -1, (double)n_ch * 1.0, 0.0, false, 1.0, 0.0);
                pose(6, size, 1.35, size + gap, 3.141592653589793);
            }
            WorldGen::PlaceOpt a;
            a.has_min_x = a.has_max_x = a.has_min_z = a.has_max_z = true;
            a.min_x = 4; a.max_x = 5; a.min_z = 4; a.max_z = 6;
            place_agent(a, arad);
        } else if (TASK_ == MWB_TASK_SIDEWALK) {   // envs/sidewalk.py:22-72
            w.add_rect_room_ex(-3, 0, 0, 12, TEXF_BRICK_WALL, TEXF_CONCRETE_TILES, TEXF_CONCRETE_TILES);
            w.add_rect_room_ex(0, 6, -80, 80, TEXF_CONCRETE, TEXF_ASPHALT, TEXF_CONCRETE_TILES);
            w.connect_rooms(0, 1, 2, 0, 12, false, 0);
            w.gen_static_data(dr, d.S_max);
            put_mesh(0, MWB_MESH_BUILDING, 30, true, -1); place_at(0, 30, 0.0, 30, true, -3.141592653589793);
            for (int i = 1; i < 6; i++) { put_mesh(i, MWB_MESH_CONE, 0.75, true, -1); place_at(i, 1, 0.0, 2 * i, false, 0); }
            put_box(6, 0.8, 4);
            WorldGen::PlaceOpt o;
            o.room = 0; o.has_min_z = o.has_max_z = true; o.min_z = rooms[0].max_z - 2; o.max_z = rooms[0].max_z;
            place(6, o);
            WorldGen::PlaceOpt a;
            a.room = 0; a.has_min_z = a.has_max_z = true; a.min_z = 0; a.max_z = 1.5;
            place_agent(a, arad);
        } else {   // MWB_TASK_WALLGAP, envs/wallgap.py:21-52
            w.add_rect_room_ex(-7, 7, 0.5, 8, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES);
            w.add_rect_room_ex(-7, 7, -8, -0.5, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES);
            w.connect_rooms(0, 1, 1, -1.5, 1.5, false, 0);
            w.gen_static_data(dr, d.S_max);
            put_box(0, 0.8, 4);
            WorldGen::PlaceOpt o;
            o.room = 1;
            place(0, o);
            put_mesh(1, MWB_MESH_BUILDING, 30, true, -1); place_at(1, 30, 0.0, 30, true, -3.141592653589793);
            WorldGen::PlaceOpt a;
            a.room = 0;
            place_agent(a, arad);
        }
    } else if (TASK_ == MWB_TASK_HALLWAY) {   // envs/hallway.py:25-42
        double length = d.task_args[0];
        int r = w.add_rect_room(-1, -1 + length, -2, 2, TEXF_CONCRETE);
        double rmax = rooms[r].max_x;
        w.gen_static_data(dr, d.S_max);
        w.place_entity(brad, false, 0, 0, 0, false, 0, true, rmax - 2, false, 0, box_x, box_z, box_dir);
        double adir = w.rng.uniform(-3.141592653589793 / 4, 3.141592653589793 / 4);
        w.place_entity(arad, true, box_x, box_z, brad, true, adir, false, 0, true, rmax - 2, ag_x, ag_z, ag_dir);
    } else if (TASK_ == MWB_TASK_ONEROOM) {   // envs/oneroom.py:26-35
        double size = d.task_args[0];
        w.add_rect_room(0, size, 0, size, TEXF_CONCRETE);
        w.gen_static_data(dr, d.S_max);
        w.place_entity(brad, false, 0, 0, 0, false, 0, false, 0, false, 0, box_x, box_z, box_dir);
        w.place_entity(arad, true, box_x, box_z, brad, false, 0, false, 0, false, 0, ag_x, ag_z, ag_dir);
    } else if (TASK_ == MWB_TASK_FOURROOMS) {   // envs/fourrooms.py:22-52
        w.add_rect_room(-7, -1, 1, 7, TEXF_CONCRETE);
        w.add_rect_room(1, 7, 1, 7, TEXF_CONCRETE);
        w.add_rect_room(1, 7, -7, -1, TEXF_CONCRETE);
        w.add_rect_room(-7, -1, -7, -1, TEXF_CONCRETE);
        w.connect_rooms(0, 1, 2, 3, 5, true, 2.2);
        w.connect_rooms(1, 2, 1, 3, 5, true, 2.2);
        w.connect_rooms(2, 3, 2, -5, -3, true, 2.2);
        w.connect_rooms(3, 0, 1, -5, -3, true, 2.2);
        w.gen_static_data(dr, d.S_max);
        w.place_entity(brad, false, 0, 0, 0, false, 0, false, 0, false, 0, box_x, box_z, box_dir);
        w.place_entity(arad, true, box_x, box_z, brad, false, 0, false, 0, false, 0, ag_x, ag_z, ag_dir);
    } else {   // envs/maze.py:34-104, recursive backtracker with an explicit stack
        int num_rows = (int)d.task_args[0], num_cols = (int)d.task_args[1];
        double room_size = d.task_args[2], gap = 0.25;
        for (int j = 0; j < num_rows; j++)
            for (int i = 0; i < num_cols; i++) {
                double min_x = i * (room_size + gap), max_x = min_x + room_size;
                double min_z = j * (room_size + gap), max_z = min_z + room_size;
                w.add_rect_room(min_x, max_x, min_z, max_z, TEXF_BRICK_WALL);
            }
        // DFS frames: cell, packed neighbour order, next index
        lds_i32 *stack = dfs;
        LDS_AS uint8_t *visited = (LDS_AS uint8_t *)(stack + 3 * (num_rows * num_cols + 1));
        for (int i = 0; i < num_rows * num_cols; i++) visited[i] = 0;
        __syncthreads();
        int sp = 0;
        auto push = [&](int ci, int cj) {
            visited[cj * num_cols + ci] = 1;
            // RandGen.subset(list, 4): choice + remove, random.py:50-65; entries are (dj, di)
            int lst[4] = {0, 1, 2, 3}, n = 4, packed = 0;
            for (int q = 0; q < 4; q++) {
                int idx = w.rng.randint(0, n);
                int v = idx == 0 ? lst[0] : idx == 1 ? lst[1] : idx == 2 ? lst[2] : lst[3];
                packed |= v << (2 * q);
                if (idx <= 0) lst[0] = lst[1];
                if (idx <= 1) lst[1] = lst[2];
                if (idx <= 2) lst[2] = lst[3];
                n--;
            }
            stack[sp * 3 + 0] = cj * num_cols + ci; stack[sp * 3 + 1] = packed; stack[sp * 3 + 2] = 0;
            sp++;
        };
        push(0, 0);
        while (sp > 0 && !w.fail) {
            int cell = stack[(sp - 1) * 3], packed = stack[(sp - 1) * 3 + 1], next = stack[(sp - 1) * 3 + 2];
            if (next >= 4) { sp--; continue; }
            stack[(sp - 1) * 3 + 2] = next + 1;
            int v = (packed >> (2 * next)) & 3;   // (0,1),(0,-1),(-1,0),(1,0) as (dj, di)
            int dj = v == 0 ? 0 : v == 1 ? 0 : v == 2 ? -1 : 1;
            int di = v == 0 ? 1 : v == 1 ? -1 : 0;
            int ci = cell % num_cols, cj = cell / num_cols;
            int ni = ci + di, nj = cj + dj;
            if (nj < 0 || nj >= num_rows || ni < 0 || ni >= num_cols) continue;
            if (visited[nj * num_cols + ni]) continue;
            int rb = nj * num_cols + ni;
            if (di == 0) w.connect_rooms(cell, rb, 1, rooms[cell].min_x, rooms[cell].max_x, false, 0);
            else w.connect_rooms(cell, rb, 2, rooms[cell].min_z, rooms[cell].max_z, false, 0);
            push(ni, nj);
        }
        __syncthreads();
        w.gen_static_data(dr, d.S_max);
        w.place_entity(brad, false, 0, 0, 0, false, 0, false, 0, false, 0, box_x, box_z, box_dir);
        w.place_entity(arad, true, box_x, box_z, brad, false, 0, false, 0, false, 0, ag_x, ag_z, ag_dir);
    }
    // miniworld.py:561-573: sky / light, then entity.randomize in list order (box, agent)
    double sky[3], lpos[3], lcol[3], lamb[3], cam[4];
    double biases[MWB_MAX_BOXES][3];
    sample_param(w.rng, d.params[MWB_P_SKY_COLOR], 3, dr, sky);
    sample_param(w.rng, d.params[MWB_P_LIGHT_POS], 3, dr, lpos);
    sample_param(w.rng, d.params[MWB_P_LIGHT_COLOR], 3, dr, lcol);
    sample_param(w.rng, d.params[MWB_P_LIGHT_AMBIENT], 3, dr, lamb);
    if (!(TASK_ >= MWB_TASK_PICKUPOBJS)) {
        for (int b = 0; b < d.n_boxes; b++) sample_param(w.rng, d.params[MWB_P_OBJ_COLOR_BIAS], 3, dr, biases[b]);   // entities randomize in list order
    } else {
        for (int b = 0; b < d.n_boxes; b++) {   // Box.randomize draws its colour bias, MeshEnt / ImageFrame draw nothing, TextFrame a texture per character
            const int kind = E_meta[b] & 15;
            if (kind == MWB_ENT_BOX) {
                double bias[3];
                sample_param(w.rng, d.params[MWB_P_OBJ_COLOR_BIAS], 3, dr, bias);
                __syncthreads();
                if (lane == 0) { E_bias[b * 3] = bias[0]; E_bias[b * 3 + 1] = bias[1]; E_bias[b * 3 + 2] = bias[2]; }
                __syncthreads();
            } else if (kind == MWB_ENT_TEXT) {   // TextFrame.randomize, entity.py:268-278: Texture.get('chars/ch_0x<ord>', rng) -> rng.int(0, 9)
                const int ci = (int)d.task_args[1];
                const int n_ch = ci == 0 ? 4 : ci == 1 ? 3 : 5;
                for (int c = 0; c < 8; c++) {
                    // index into "BLUERDGN" of character c of "BLUE" / "RED" / "GREEN"
                    const int ch = ci == 0 ? (c == 0 ? 0 : c == 1 ? 1 : c == 2 ? 2 : 3) : ci == 1 ? (c == 0 ? 4 : c == 1 ? 3 : 5) : (c == 0 ? 6 : c == 1 ? 4 : c == 2 ? 3 : c == 3 ? 3 : 7);
                    int t = -1;
                    if (c < n_ch) t = MWB_TEX_CHAR0 + 9 * ch + (dr ? w.rng.randint(0, 9) : 0);
                    __syncthreads();
                    if (lane == 0) E_text[c] = t;
                    __syncthreads();
                }
            }
        }
    }
    sample_param(w.rng, d.params[MWB_P_CAM_HEIGHT], 1, dr, &cam[0]);
    sample_param(w.rng, d.params[MWB_P_CAM_FWD_DISP], 1, dr, &cam[1]);
    sample_param(w.rng, d.params[MWB_P_CAM_PITCH], 1, dr, &cam[2]);
    sample_param(w.rng, d.params[MWB_P_CAM_FOV_Y], 1, dr, &cam[3]);
    __syncthreads();

    // ---- emit: RNG state, scalars (lane 0), room + segment tables (all lanes)
    for (int i = lane; i < 624; i += WAVE) st[i] = key[i];
    if (lane == 0) {
        st[624] = (uint32_t)w.rng.pos;
        d.agent_x[e] = ag_x; d.agent_z[e] = ag_z; d.agent_dir[e] = ag_dir;
        // COLORS in COLOR_NAMES order (entity.py:8-18): blue green grey purple red yellow; Box.randomize adds the bias and clips
        const double COLORS[6][3] = {{0.0, 0.0, 1.0}, {0.0, 1.0, 0.0}, {0.39, 0.39, 0.39}, {0.44, 0.15, 0.76}, {1.0, 0.0, 0.0}, {1.00, 1.00, 0.00}};
        const size_t N = (size_t)d.N;
        for (int b = 0; b < d.n_boxes && (TASK_ >= MWB_TASK_PICKUPOBJS); b++) {   // entity tasks: the slots as recorded in LDS
            const size_t be = (size_t)b * N + e;
            const int kind = E_meta[b] & 15, ci = E_col[b];
            d.box_x[be] = E_x[b]; d.box_y[be] = E_y[b]; d.box_z[be] = E_z[b]; d.box_dir[be] = E_dir[b]; d.box_size[be] = E_size[b];
            d.ent_meta[be] = E_meta[b]; d.ent_radius[be] = E_rad[b]; d.ent_height[be] = E_hgt[b]; d.ent_scale[be] = E_scale[b];
            for (int k = 0; k < 3; k++) {   // Box: COLORS[c] + bias, clipped; mesh: the material's Kd = COLORS[c], white without one
                double v = ci >= 0 ? COLORS[ci][k] : 1.0;
                if (kind == MWB_ENT_BOX) { v = v + E_bias[b * 3 + k]; v = v < 0 ? 0 : (v > 1 ? 1 : v); }
                d.box_color[be * 3 + k] = v;
            }
        }
        if ((TASK_ >= MWB_TASK_PICKUPOBJS)) {
            uint8_t *ord = d.ent_order + (size_t)e * MWB_ORDER_STRIDE;
            for (int k = 0; k < MWB_ORDER_STRIDE; k++) ord[k] = k < d.n_boxes ? (uint8_t)k : (uint8_t)MWB_ENT_AGENT;   // [0 .. E-1, agent]
            d.n_order[e] = d.n_boxes + 1;
            d.task_f[e] = task_f0; d.task_i[e] = 0; d.ovr_slot[e] = -1;
            for (int c = 0; c < 8; c++) d.text_tex[e * 8 + c] = TASK_ == MWB_TASK_SIGN ? E_text[c] : -1;
        }
        for (int b = 0; b < d.n_boxes && !(TASK_ >= MWB_TASK_PICKUPOBJS); b++) {
            int base = 4;   // red
            double x = box_x, z = box_z, dir = box_dir, sz = box_s;
            if (TASK_ == MWB_TASK_PUTNEXT) { base = b; x = pn_x[b]; z = pn_z[b]; dir = pn_dir[b]; sz = pn_s[b]; }
            else if (b == 1) { base = TASK_ == MWB_TASK_SIM2REAL_PUSH ? 5 : 0; x = box2_x; z = box2_z; dir = box2_dir; sz = box2_s; }   // yellow / blue
            const size_t be = (size_t)b * N + e;
            d.box_x[be] = x; d.box_z[be] = z; d.box_y[be] = 0.0; d.box_dir[be] = dir; d.box_size[be] = sz;
            for (int k = 0; k < 3; k++) {
                const double v = COLORS[base][k] + biases[b][k];
                d.box_color[be * 3 + k] = v < 0 ? 0 : (v > 1 ? 1 : v);
            }
        }
        for (int k = 0; k < 3; k++) {
            d.sky_color[e * 3 + k] = sky[k]; d.light_pos[e * 3 + k] = lpos[k];
            d.light_color[e * 3 + k] = lcol[k]; d.light_ambient[e * 3 + k] = lamb[k];
        }
        d.carrying[e] = -1;   // a fresh Agent() carries nothing (entity.py:446)
        d.goal_idx[e] = goal_idx; d.episode_count[e] = episode_count;
        d.goal_dist[e] = goal_dist;
        for (int k = 0; k < 4; k++) d.cam[e * 4 + k] = cam[k];
        d.step_count[e] = 0;
        d.n_rooms[e] = w.fail ? -1 : w.n_rooms;
        {   // extents of the floorplan (miniworld.py:576-579), float64: the top view's frame
            double mnx = rooms[0].min_x, mxx = rooms[0].max_x, mnz = rooms[0].min_z, mxz = rooms[0].max_z;
            for (int i = 1; i < w.n_rooms; i++) {
                mnx = fmin(mnx, rooms[i].min_x); mxx = fmax(mxx, rooms[i].max_x);
                mnz = fmin(mnz, rooms[i].min_z); mxz = fmax(mxz, rooms[i].max_z);
            }
            d.world_ext[e * 4 + 0] = mnx; d.world_ext[e * 4 + 1] = mxx; d.world_ext[e * 4 + 2] = mnz; d.world_ext[e * 4 + 3] = mxz;
        }
        if (w.fail) atomicExch(d.error_flag, e + 1);
        d.n_segs[e] = w.n_segs;
    }
    for (int i = lane; i < w.n_segs * 4; i += WAVE) d.segs[(size_t)i * d.N + e] = segs[i];   // transposed, see step_kernel
    float *grooms = d.rooms + (size_t)e * d.R_max * d.room_words;
    // A room whose outline runs the other way round (the connectors connect_rooms builds between YMaze's hub and arms:
    // miniworld.py:826 lists their corners clockwise there) has every polygon facing away from its inside - with back-face
    // culling on (miniworld.py:498-499) nothing of it is drawn.  Test: the "inward" normals point away from the centroid.
    auto culled = [&](const LRoom &r) {
        double cx = 0, cz = 0, inward = 0;
        for (int k = 0; k < r.ne; k++) { cx += r.ox[k] / r.ne; cz += r.oz[k] / r.ne; }
        for (int k = 0; k < r.ne; k++) {
            double nx, nz;
            w.edge_normal(r, k, nx, nz);
            inward += nx * (cx - r.ox[k]) + nz * (cz - r.oz[k]);
        }
        return inward < 0;
    };
    for (int ri = lane; d.poly && ri < w.n_rooms; ri += WAVE) {   // polygon room table (mwb_internal.h)
        const LRoom &r = rooms[ri];
        float *o = grooms + (size_t)ri * MWB_POLY_ROOM_WORDS;
        const bool cull_r = culled(r);
        o[PW_HEIGHT] = d.no_ceiling ? -(float)r.height : (float)r.height;
        o[PW_TEX] = __int_as_float(r.tex_id[0] | (r.tex_id[1] << 8) | (r.tex_id[2] << 16));
        o[PW_FLAGS] = __int_as_float(r.ne | (cull_r ? 256 : 0));
        o[3] = 0.0f;
        for (int k = 0; k < 4; k++) {
            float *ed = o + PW_EDGE0 + PW_EDGE_WORDS * k;
            for (int q = 0; q < PW_EDGE_WORDS; q++) ed[q] = 0.0f;
            ed[9] = __int_as_float(-1);
            if (k >= r.ne) continue;
            double nx, nz;
            w.edge_normal(r, k, nx, nz);
            double edx_, edz_, elen_;
            w.edge(r, k, edx_, edz_, elen_);
            ed[0] = (float)r.ox[k]; ed[1] = (float)r.oz[k]; ed[2] = (float)edx_; ed[3] = (float)edz_;
            ed[4] = (float)nx; ed[5] = (float)nz;
            if (r.n_port[k]) {
                ed[6] = (float)r.p_start[k]; ed[7] = (float)r.p_end[k]; ed[8] = (float)r.p_maxy[k];
                int nb = r.nbr[k];
                if (nb >= 0 && !cull_r && culled(rooms[nb])) {   // by-pass a culled connector: on to the room behind its other portal
                    const LRoom &c = rooms[nb];
                    int onward = -1;
                    for (int q = 0; q < c.ne; q++)
                        if (c.n_port[q] == 1 && c.nbr[q] != ri) onward = c.nbr[q];
                    if (onward < 0 || culled(rooms[onward])) atomicExch(d.error_flag, e + 1);
                    nb = onward;
                }
                if (nb < 0) atomicExch(d.error_flag, e + 1);
                ed[9] = __int_as_float(nb);
            }
        }
    }
    // A wall with TWO openings (ThreeRooms) does not fit a room record with one portal per side: the room is cut into two RENDER
    // rooms along the line midway between the openings, joined by a "virtual" portal over the whole cut (nothing is drawn there;
    // a convex room cut by a plane is two convex rooms, visibility is unchanged and every surface keeps its texture origin).  The
    // half at the lower coordinate keeps the room's index, the other one is appended behind the last room.
    int cut_room = -1, cut_s = 0, n_rrooms = w.n_rooms;
    float cut = 0.0f, c_first[3] = {0, 0, 0}, c_second[3] = {0, 0, 0};   // lo hi max_y of the opening at the lower / higher coordinate
    int c_nbr[2] = {-1, -1};
    if (!d.poly && w.xp_room >= 0) {
        const LRoom &r = rooms[w.xp_room];
        const int ed = w.xp_edge;
        double dx, dz, elen_;
        w.edge(r, ed, dx, dz, elen_);
        const double nx = dz, nz = -dx;
        cut_room = w.xp_room;
        cut_s = (nx == -1 && nz == 0) ? 0 : (nx == 0 && nz == 1) ? 1 : (nx == 1 && nz == 0) ? 2 : 3;
        const bool along_z = (cut_s == 0 || cut_s == 2);
        const double p0c = along_z ? r.oz[ed] : r.ox[ed], dirc = along_z ? dz : dx;
        const double a0 = p0c + dirc * r.p_start[ed], a1 = p0c + dirc * r.p_end[ed], b0 = p0c + dirc * w.xp_start, b1 = p0c + dirc * w.xp_end;
        float pa[3] = {(float)(a0 < a1 ? a0 : a1), (float)(a0 < a1 ? a1 : a0), (float)r.p_maxy[ed]};
        float pb[3] = {(float)(b0 < b1 ? b0 : b1), (float)(b0 < b1 ? b1 : b0), (float)w.xp_maxy};
        const bool swap = pb[0] < pa[0];
        for (int k = 0; k < 3; k++) { c_first[k] = swap ? pb[k] : pa[k]; c_second[k] = swap ? pa[k] : pb[k]; }
        c_nbr[0] = swap ? w.xp_nbr : r.nbr[ed]; c_nbr[1] = swap ? r.nbr[ed] : w.xp_nbr;
        cut = (float)(0.5 * ((double)c_first[1] + (double)c_second[0]));
        n_rrooms = w.n_rooms + 1;
        if (r.n_port[0] + r.n_port[1] + r.n_port[2] + r.n_port[3] > 2 || n_rrooms > d.R_max || c_nbr[0] < 0 || c_nbr[1] < 0) {
            atomicExch(d.error_flag, e + 1);
            cut_room = -1; n_rrooms = w.n_rooms;
        }
    }
    if (lane == 0) d.n_rrooms[e] = w.fail ? -1 : n_rrooms;
    for (int ri = lane; !d.poly && ri < w.n_rooms; ri += WAVE) {
        const LRoom &r = rooms[ri];
        float *o = grooms + (size_t)ri * MWB_ROOM_WORDS;
        o[RW_MINX] = (float)r.min_x; o[RW_MAXX] = (float)r.max_x; o[RW_MINZ] = (float)r.min_z; o[RW_MAXZ] = (float)r.max_z;
        o[RW_HEIGHT] = d.no_ceiling ? -(float)r.height : (float)r.height;   // negative: no ceiling polygon (trace_rooms)
        uint32_t texw = (uint32_t)(r.tex_id[0] | (r.tex_id[1] << 8) | (r.tex_id[2] << 16));
        uint32_t nbrs[4] = {RW_NO_NBR, RW_NO_NBR, RW_NO_NBR, RW_NO_NBR};
        for (int s = 0; s < 4; s++) {
            float *sd = o + RW_SIDE0 + RW_SIDE_WORDS * s;
            sd[RS_LO] = 0; sd[RS_HI] = 0; sd[RS_MAXY] = 0; sd[RS_UORG] = 0;
        }
        for (int ed = 0; ed < 4; ed++) {
            double dx, dz, len;
            w.edge(r, ed, dx, dz, len);
            double nx = dz, nz = -dx;   // inward normal
            int s = (nx == -1 && nz == 0) ? 0 : (nx == 0 && nz == 1) ? 1 : (nx == 1 && nz == 0) ? 2 : 3;
            bool along_z = (s == 0 || s == 2);
            double p0c = along_z ? r.oz[ed] : r.ox[ed], dirc = along_z ? dz : dx;
            float *sd = o + RW_SIDE0 + RW_SIDE_WORDS * s;
            sd[RS_UORG] = (float)p0c;
            if (dirc < 0) texw |= 1u << (24 + s);
            if (r.n_port[ed]) {
                double c0 = p0c + dirc * r.p_start[ed], c1 = p0c + dirc * r.p_end[ed];
                sd[RS_LO] = (float)(c0 < c1 ? c0 : c1); sd[RS_HI] = (float)(c0 < c1 ? c1 : c0);
                sd[RS_MAXY] = (float)r.p_maxy[ed];
                int nb = r.nbr[ed];
                if (nb == cut_room && cut_room >= 0 && sd[RS_LO] >= cut) nb = w.n_rooms;   // the opening leads into the appended half
                nbrs[s] = (uint32_t)nb & 0xFFFFu;
            }
        }
        if (ri == cut_room) {
            const bool ax0 = (cut_s == 1 || cut_s == 3);   // the cut wall runs along x
            const int lo_end = ax0 ? 2 : 1, hi_end = ax0 ? 0 : 3;   // the sides at the low / high end of that axis
            float *h = grooms + (size_t)w.n_rooms * MWB_ROOM_WORDS;
            for (int k = 0; k < MWB_ROOM_WORDS; k++) h[k] = o[k];
            o[ax0 ? RW_MAXX : RW_MAXZ] = cut; h[ax0 ? RW_MINX : RW_MINZ] = cut;
            float *so = o + RW_SIDE0 + RW_SIDE_WORDS * cut_s, *sh = h + RW_SIDE0 + RW_SIDE_WORDS * cut_s;
            so[RS_LO] = c_first[0]; so[RS_HI] = c_first[1]; so[RS_MAXY] = c_first[2];
            sh[RS_LO] = c_second[0]; sh[RS_HI] = c_second[1]; sh[RS_MAXY] = c_second[2];
            float *vo = o + RW_SIDE0 + RW_SIDE_WORDS * hi_end, *vh = h + RW_SIDE0 + RW_SIDE_WORDS * lo_end;
            vo[RS_LO] = vh[RS_LO] = (ax0 ? (float)r.min_z : (float)r.min_x) - 1.0f;
            vo[RS_HI] = vh[RS_HI] = (ax0 ? (float)r.max_z : (float)r.max_x) + 1.0f;
            vo[RS_MAXY] = vh[RS_MAXY] = 1e30f; vo[RS_UORG] = vh[RS_UORG] = 0.0f;
            uint32_t nh[4] = {nbrs[0], nbrs[1], nbrs[2], nbrs[3]};
            nbrs[cut_s] = (uint32_t)c_nbr[0] & 0xFFFFu; nh[cut_s] = (uint32_t)c_nbr[1] & 0xFFFFu;
            nbrs[hi_end] = (uint32_t)w.n_rooms; nh[lo_end] = (uint32_t)ri;
            h[RW_TEX] = __int_as_float((int)(texw & ~(1u << (24 + lo_end))));
            texw &= ~(1u << (24 + hi_end));
            h[RW_NBR01] = __int_as_float((int)(nh[0] | (nh[1] << 16)));
            h[RW_NBR23] = __int_as_float((int)(nh[2] | (nh[3] << 16)));
        }
        o[RW_TEX] = __int_as_float((int)texw);
        o[RW_NBR01] = __int_as_float((int)(nbrs[0] | (nbrs[1] << 16)));
        o[RW_NBR23] = __int_as_float((int)(nbrs[2] | (nbrs[3] << 16)));
    }
    // the frame constants of the new episode's first view, right here (lane 0 wrote the state it reads): the side stream
    // needs no prep launch of its own between this kernel and the render of the regenerated envs
    if (lane == 0) prep_env(d, e);
    }   // env loop
}

// ================================================================================== prep kernel
// Camera basis (Agent.cam_pos / cam_dir, entity.py:457-484 via gen_rot_matrix math.py:9-23), the
// gluPerspective / gluLookAt set-up of render_obs (miniworld.py:1183-1200), fixed-function lighting
// per flat face (miniworld.py:1026-1045) and the box frame (entity.py:385-408), once per env-step.
__device__ __forceinline__ void lit_color(const float *L, const float *amb, const float *dif, float nx, float ny, float nz,
                                          const float *C, float *out) {
    float ndl = nx * L[0] + ny * L[1] + nz * L[2];
    if (ndl < 0) ndl = 0;
    for (int k = 0; k < 3; k++) {
        float v = (0.2f * C[k] + amb[k] * C[k]) + ndl * dif[k] * C[k];   // 0.2 = GL default global ambient
        out[k] = v > 1.0f ? 1.0f : v;
    }
}

__device__ __forceinline__ void prep_env(const MwbDev &d, int e) {
    float *fc = d.frame + (size_t)e * d.frame_words;
    double adir = d.agent_dir[e];
    double cam_h = d.cam[e * 4 + 0], cam_fd = d.cam[e * 4 + 1], cam_pitch = d.cam[e * 4 + 2], fov = d.cam[e * 4 + 3];
    double rot_y[9], rot_z[9], disp[3], t[3], cd[3];
    rot_matrix(0, 1, 0, adir, rot_y);
    double v0[3] = {cam_fd, cam_h, 0};
    vec_mat(v0, rot_y, disp);
    double cp[3] = {d.agent_x[e] + disp[0], 0.0 + disp[1], d.agent_z[e] + disp[2]};
    rot_matrix(0, 0, 1, cam_pitch * 3.141592653589793 / 180, rot_z);
    double xv[3] = {1, 0, 0};
    vec_mat(xv, rot_z, t);
    vec_mat(t, rot_y, cd);
    double fl = sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]);
    double f[3] = {cd[0] / fl, cd[1] / fl, cd[2] / fl};
    double sl = sqrt(f[2] * f[2] + f[0] * f[0]);
    double s[3] = {-f[2] / sl, 0, f[0] / sl};
    double u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    double th = tan(fov * 3.141592653589793 / 180 / 2);
    float eye[3];
    for (int k = 0; k < 3; k++) {
        eye[k] = (float)cp[k];
        fc[FC_EYE + k] = eye[k]; fc[FC_F + k] = (float)f[k]; fc[FC_S + k] = (float)s[k]; fc[FC_U + k] = (float)u[k];
    }
    fc[FC_TH] = (float)th;
    fc[FC_TW] = (float)(th * ((double)d.W / (double)d.H));
    // light: GL_POSITION = (light_pos + 1, w = 0) -> directional
    double lp[3] = {d.light_pos[e * 3] + 1, d.light_pos[e * 3 + 1] + 1, d.light_pos[e * 3 + 2] + 1};
    double ll = sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]);
    float L[3], amb[3], dif[3];
    for (int k = 0; k < 3; k++) {
        L[k] = (float)(lp[k] / ll); amb[k] = (float)d.light_ambient[e * 3 + k]; dif[k] = (float)d.light_color[e * 3 + k];
        fc[FC_SKY + k] = (float)d.sky_color[e * 3 + k];
    }
    const float white[3] = {1, 1, 1};
    lit_color(L, amb, dif, 0, 1, 0, white, fc + FC_LIT_FLOOR);
    lit_color(L, amb, dif, 0, -1, 0, white, fc + FC_LIT_CEIL);
    lit_color(L, amb, dif, -1, 0, 0, white, fc + FC_LIT_WALL + 0);
    lit_color(L, amb, dif, 0, 0, 1, white, fc + FC_LIT_WALL + 3);
    lit_color(L, amb, dif, 1, 0, 0, white, fc + FC_LIT_WALL + 6);
    lit_color(L, amb, dif, 0, 0, -1, white, fc + FC_LIT_WALL + 9);
    if (d.poly)   // walls of any orientation: the light itself, the colour is computed per edge in the render kernel
        for (int k = 0; k < 3; k++) { fc[FC_LIGHT_DIR + k] = L[k]; fc[FC_LIGHT_AMB + k] = amb[k]; fc[FC_LIGHT_DIF + k] = dif[k]; }
    // Frame-level gate for every box test of the render kernel: can any box's (pixel-footprint-inflated) bounding sphere
    // meet the cone around the view direction that contains the whole image (half angle atan |(TW, TH)|)?  Conservative
    // (2 % + 0.01 rad slack); in a maze the box is out of view in most frames and the tests are skipped wholesale.
    float any_in_view = 0.0f;
    for (int bi = 0; bi < (d.ent_task ? MWB_MAX_ENTS : d.n_boxes); bi++) {   // one block of FC_BOX_STRIDE words per box / entity slot
    fc = d.frame + (size_t)e * d.frame_words + bi * FC_BOX_STRIDE;
    if (bi >= d.n_boxes) {   // entity tasks: the render kernel is compiled for MWB_MAX_ENTS slots; the unused ones meet no ray
        fc[FC_BOX_HX] = -3.0f;
        fc[FC_CULL_OC] = 0.0f; fc[FC_CULL_OC + 1] = 0.0f; fc[FC_CULL_OC + 2] = 0.0f;
        fc[FC_CULL_CC] = INFINITY; fc[FC_CULL_CC_PIXEL] = INFINITY;
        continue;
    }
    const size_t be = (size_t)bi * d.N + e;
    float bcol[3];
    for (int k = 0; k < 3; k++) bcol[k] = (float)d.box_color[be * 3 + k];
    double bdir = d.box_dir[be];
    double px = d.box_x[be], py = d.box_y[be], pz = d.box_z[be];
    if (d.ent_task) {   // entity tasks: the block is read by kind (mwb_internal.h); an entity that left the list is never met by a ray
        const int meta = d.ent_meta[be];
        const int kind = MWB_META_KIND(meta);
        bool alive = MWB_META_ALIVE(meta) != 0;
        if (d.step_pass && d.ovr_slot[e] == bi) {   // the step's own frame: as the entity was before the task rule removed / respawned it
            px = d.ovr_pose[e * 4 + 0]; py = d.ovr_pose[e * 4 + 1]; pz = d.ovr_pose[e * 4 + 2]; bdir = d.ovr_pose[e * 4 + 3];
            alive = true;
        }
        float *blk = fc + FC_LIT_BOX;
        const float bpos[3] = {(float)px, (float)py, (float)pz};
        const float c_ = (float)ref_cos(bdir), s_ = (float)ref_sin(bdir);
        fc[FC_BOX_POS] = bpos[0]; fc[FC_BOX_POS + 1] = bpos[1]; fc[FC_BOX_POS + 2] = bpos[2];
        fc[FC_BOX_C] = c_; fc[FC_BOX_S] = s_;
        const float ro[3] = {eye[0] - bpos[0], eye[1] - bpos[1], eye[2] - bpos[2]};
        float sph_c[3] = {0, 0, 0}, R = 0.0f;   // bounding sphere (world), 2 % larger
        if (!alive) {
            fc[FC_BOX_HX] = -3.0f;
            fc[FC_CULL_OC] = 0.0f; fc[FC_CULL_OC + 1] = 0.0f; fc[FC_CULL_OC + 2] = 0.0f;
            fc[FC_CULL_CC] = INFINITY; fc[FC_CULL_CC_PIXEL] = INFINITY;
            continue;
        }
        if (kind == MWB_ENT_MESH) {
            const MwbMeshDesc &md = d.mesh_desc[MWB_META_GEOM(meta)];
            const double scale = d.ent_scale[be];
            const float inv_s = (float)(1.0 / scale);
            fc[FC_BOX_HX] = -1.0f; fc[FC_BOX_HZ] = 0.0f;
            blk[FE_MESH_LL] = (L[0] * c_ - L[2] * s_) * inv_s; blk[FE_MESH_LL + 1] = L[1] * inv_s; blk[FE_MESH_LL + 2] = (L[0] * s_ + L[2] * c_) * inv_s;
            for (int k = 0; k < 3; k++) { blk[FE_MESH_KD + k] = bcol[k]; blk[FE_MESH_AMB + k] = 0.2f * bcol[k] + amb[k] * bcol[k]; blk[FE_MESH_DIF + k] = dif[k]; }
            blk[FE_MESH_INVS] = inv_s; blk[FE_MESH_GEOM] = __int_as_float(MWB_META_GEOM(meta)); blk[FE_MESH_TEX] = __int_as_float(md.tex_id);
            blk[FE_MESH_BHX] = fmaxf(fabsf(md.min_c[0]), fabsf(md.max_c[0])); blk[FE_MESH_BHZ] = fmaxf(fabsf(md.min_c[2]), fabsf(md.max_c[2]));
            fc[FC_BOX_SY] = md.max_c[1];
            fc[FC_BOX_LO] = (ro[0] * c_ - ro[2] * s_) * inv_s; fc[FC_BOX_LO + 1] = ro[1] * inv_s; fc[FC_BOX_LO + 2] = (ro[0] * s_ + ro[2] * c_) * inv_s;
            float lc[3], ext2 = 0.0f;
            for (int k = 0; k < 3; k++) { lc[k] = 0.5f * (md.min_c[k] + md.max_c[k]); const float h = 0.5f * (md.max_c[k] - md.min_c[k]); ext2 += h * h; }
            const float sc = (float)scale;
            sph_c[0] = bpos[0] + sc * (lc[0] * c_ + lc[2] * s_); sph_c[1] = bpos[1] + sc * lc[1]; sph_c[2] = bpos[2] + sc * (-lc[0] * s_ + lc[2] * c_);
            R = 1.02f * sc * sqrtf(ext2);
        } else if (kind == MWB_ENT_IMAGE || kind == MWB_ENT_TEXT) {
            const float sx = 0.05f, hy = (float)(d.ent_height[be] / 2), hz = (float)(d.box_size[be] / 2);   // depth, height / 2, width / 2
            fc[FC_BOX_HX] = -2.0f; fc[FC_BOX_HZ] = 0.0f; fc[FC_BOX_SY] = 0.0f;
            lit_color(L, amb, dif, c_, 0.0f, -s_, white, blk + FE_FRAME_LIT);   // the front's normal (1, 0, 0) turned by the heading
            blk[FE_FRAME_SX] = sx; blk[FE_FRAME_HY] = hy; blk[FE_FRAME_HZ] = hz;
            blk[FE_FRAME_CW] = kind == MWB_ENT_TEXT ? (float)d.ent_height[be] : (float)d.box_size[be];
            int n_ch = 1;
            if (kind == MWB_ENT_TEXT) { n_ch = 0; for (int c = 0; c < 8; c++) if (d.text_tex[e * 8 + c] >= 0) n_ch = c + 1; }
            blk[FE_FRAME_NCH] = __int_as_float(n_ch);
            for (int c = 0; c < 8; c++) blk[FE_FRAME_TEX + c] = __int_as_float(kind == MWB_ENT_TEXT ? d.text_tex[e * 8 + c] : (c == 0 ? 20 : -1));   // 20: logo_mila_1
            fc[FC_BOX_LO] = ro[0] * c_ - ro[2] * s_; fc[FC_BOX_LO + 1] = ro[1]; fc[FC_BOX_LO + 2] = ro[0] * s_ + ro[2] * c_;
            sph_c[0] = bpos[0] + 0.5f * sx * c_; sph_c[1] = bpos[1]; sph_c[2] = bpos[2] - 0.5f * sx * s_;
            R = 1.02f * sqrtf(0.25f * sx * sx + hy * hy + hz * hz);
        }
        if (kind != MWB_ENT_BOX) {   // cull sphere, its pixel-inflated twin and the frame-level gate, as for a box below
            const float oc[3] = {sph_c[0] - eye[0], sph_c[1] - eye[1], sph_c[2] - eye[2]};
            fc[FC_CULL_OC] = oc[0]; fc[FC_CULL_OC + 1] = oc[1]; fc[FC_CULL_OC + 2] = oc[2];
            const float oc2 = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2];
            fc[FC_CULL_CC] = oc2 - R * R;
            const float tw = (float)(th * ((double)d.W / (double)d.H)) / (float)d.W, thh = (float)th / (float)d.H;
            const float rho = 1.5f * 2.0f * sqrtf(tw * tw + thh * thh);
            const float Rp = R + rho * (sqrtf(oc2) + R);
            fc[FC_CULL_CC_PIXEL] = oc2 - Rp * Rp;
            // a mesh's own box (a sphere is a poor bound of a 30 m building): grown by the same pixel footprint, in mesh units
            if (kind == MWB_ENT_MESH) blk[FE_MESH_BPAD] = 1.02f * rho * (sqrtf(oc2) + R) * blk[FE_MESH_INVS] + 1e-3f * fc[FC_BOX_SY];
            const double dist = sqrt((double)oc2), Rg = 1.02 * (double)Rp;
            bool in_view = dist <= Rg;
            if (!in_view) {
                const double cosang = ((double)oc[0] * f[0] + (double)oc[1] * f[1] + (double)oc[2] * f[2]) / dist;
                const double ang = acos(cosang < -1 ? -1 : (cosang > 1 ? 1 : cosang));
                const double half = atan(sqrt(th * th * ((double)d.W / d.H) * ((double)d.W / d.H) + th * th));
                in_view = ang <= half + asin(Rg / dist) + 0.01;
            }
            if (in_view) any_in_view = 1.0f;
            if (kind == MWB_ENT_MESH) fc[FC_BOX_HZ] = in_view ? 1.0f : 0.0f;   // read by the render kernel's decision to stage the mesh in LDS
            continue;
        }
    }
    float bc = (float)ref_cos(bdir), bs = (float)ref_sin(bdir);
    const float ln[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int k = 0; k < 6; k++)   // world normal = R_y(dir) n_local (glRotatef about +Y)
        lit_color(L, amb, dif, ln[k][0] * bc + ln[k][2] * bs, ln[k][1], -ln[k][0] * bs + ln[k][2] * bc, bcol, fc + FC_LIT_BOX + 3 * k);
    float bpos[3] = {(float)px, (float)py, (float)pz};   // y > 0 while the box is carried
    const double bsz = d.box_size[be];   // Box.render: extents +-sx/2, 0..sy, +-sz/2 (entity.py:385-408)
    float hx = (float)(bsz / 2), hz = (float)(bsz / 2), sy = (float)bsz;
    fc[FC_BOX_POS] = bpos[0]; fc[FC_BOX_POS + 1] = bpos[1]; fc[FC_BOX_POS + 2] = bpos[2];
    fc[FC_BOX_C] = bc; fc[FC_BOX_S] = bs; fc[FC_BOX_HX] = hx; fc[FC_BOX_HZ] = hz; fc[FC_BOX_SY] = sy;
    float ro[3] = {eye[0] - bpos[0], eye[1] - bpos[1], eye[2] - bpos[2]};
    fc[FC_BOX_LO] = ro[0] * bc - ro[2] * bs; fc[FC_BOX_LO + 1] = ro[1]; fc[FC_BOX_LO + 2] = ro[0] * bs + ro[2] * bc;
    // conservative bounding sphere of the box for ray culling (radius inflated 2 %)
    float oc[3] = {bpos[0] - eye[0], (bpos[1] + 0.5f * sy) - eye[1], bpos[2] - eye[2]};
    float R = 1.02f * sqrtf(hx * hx + 0.25f * sy * sy + hz * hz);
    fc[FC_CULL_OC] = oc[0]; fc[FC_CULL_OC + 1] = oc[1]; fc[FC_CULL_OC + 2] = oc[2];
    float oc2 = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2];
    fc[FC_CULL_CC] = oc2 - R * R;
    // the same sphere grown by a pixel's footprint at the far side of the box: any ray within 1.5 pixel
    // half-diagonals of a pixel's centre ray that touches the box has its centre ray inside this one
    float tw = (float)(th * ((double)d.W / (double)d.H)) / (float)d.W, thh = (float)th / (float)d.H;
    float rho = 1.5f * 2.0f * sqrtf(tw * tw + thh * thh);
    float Rp = R + rho * (sqrtf(oc2) + R);
    fc[FC_CULL_CC_PIXEL] = oc2 - Rp * Rp;
    {
        const double dist = sqrt((double)oc2), Rg = 1.02 * (double)Rp;
        if (dist <= Rg) any_in_view = 1.0f;
        else {
            const double cosang = ((double)oc[0] * f[0] + (double)oc[1] * f[1] + (double)oc[2] * f[2]) / dist;
            const double ang = acos(cosang < -1 ? -1 : (cosang > 1 ? 1 : cosang));
            const double half = atan(sqrt(th * th * ((double)d.W / d.H) * ((double)d.W / d.H) + th * th));
            if (ang <= half + asin(Rg / dist) + 0.01) any_in_view = 1.0f;
        }
    }
    }   // boxes
    (d.frame + (size_t)e * d.frame_words)[FC_BOX_IN_VIEW] = any_in_view;
}

// Counting sort of the envs by the measured render cost of a recent frame, most expensive first (256 buckets of
// 1.28 us): the bulk render's blockIdx -> env map for the NEXT step.  One block, off the critical path (side stream);
// it may read costs the current bulk render is just rewriting - either value will do, but each env's bucket is
// decided once (kept in d.bucket) so that the result is a permutation whatever changes underneath.  The order
// inside a bucket is whatever the atomics give - results do not depend on it.
__global__ void __launch_bounds__(1024) order_kernel(MwbDev d) {
    int32_t *__restrict__ order_out = d.order_bufs[d.order_state[0] ^ 1];
    __shared__ int hist[256], scan[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < d.N; i += 1024) {
        const uint32_t c = (d.cost[2 * i] + d.cost[2 * i + 1]) >> 7;
        const int bk = 255 - (int)(c > 255u ? 255u : c);
        d.bucket[i] = (uint8_t)bk;
        atomicAdd(&hist[bk], 1);
    }
    __syncthreads();
    const int own = tid < 256 ? hist[tid] : 0;
    if (tid < 256) scan[tid] = own;
    __syncthreads();
    for (int ofs = 1; ofs < 256; ofs <<= 1) {   // inclusive prefix, 8 doubling steps
        const int v = (tid < 256 && tid >= ofs) ? scan[tid - ofs] : 0;
        __syncthreads();
        if (tid < 256) scan[tid] += v;
        __syncthreads();
    }
    if (tid < 256) hist[tid] = scan[tid] - own;
    __syncthreads();
    for (int i = tid; i < d.N; i += 1024) order_out[atomicAdd(&hist[d.bucket[i]], 1)] = i;   // own writes of pass 1
    __syncthreads();
    if (tid == 0) { __threadfence(); d.order_state[1] = 1; }
}
void mwb_launch_order(const MwbDev &d, hipStream_t s) {
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, s, d);
}

__global__ void __launch_bounds__(256) prep_kernel(MwbDev d, int mode) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 1) {
        __builtin_amdgcn_s_setprio(3);   // beside the bulk render (see reset_kernel)   // the regenerated envs, through the compact list
        const int count = d.reset_count[0];
        for (int li = e; li < count; li += gridDim.x * blockDim.x) prep_env(d, d.reset_list[li]);
        return;
    }
    if (e >= d.N) return;
    if (mode == 2 && d.reset_set[e]) return;
    prep_env(d, e);
}

// ================================================================================ render kernel
__constant__ float c_sample_x[8] = {1 * 0.0625f, -1 * 0.0625f, 5 * 0.0625f, -3 * 0.0625f, -5 * 0.0625f, -7 * 0.0625f, 3 * 0.0625f, 7 * 0.0625f};
__constant__ float c_sample_y[8] = {-3 * 0.0625f, 3 * 0.0625f, 1 * 0.0625f, -5 * 0.0625f, 5 * 0.0625f, -1 * 0.0625f, 7 * 0.0625f, -7 * 0.0625f};

#define KIND_SKY 0u
#define KIND_FLOOR 1u
#define KIND_CEIL 2u
#define KIND_WALL 3u
#define KIND_BOX 4u
#define MAKE_KEY(kind, side, room) ((uint32_t)(kind) | ((uint32_t)(side) << 3) | ((uint32_t)(room) << 6))

struct Cam {
    float eye[3], F[3], S[3], U[3], TW, TH, invW, invH, Wf, Hf;
};

// ---- visibility: these functions are part of the bit-exact render spec -------
__device__ __forceinline__ void make_ray(const Cam &c, float wx, float wy, float *dv) {
    float nx = (2.0f * wx - c.Wf) * c.invW, ny = (2.0f * wy - c.Hf) * c.invH;
    float ax = nx * c.TW, ay = ny * c.TH;
#pragma unroll
    for (int k = 0; k < 3; k++) dv[k] = fmaf(c.U[k], ay, fmaf(c.S[k], ax, c.F[k]));
}

// Portal traversal of the LDS room table: nearest opaque surface along (eye, d).
// All LDS reads of a step are issued together (both candidate sides are fetched) and the loop has
// one exit.  With PATH the sequence of sides crossed is returned as 2 bits per crossing under a leading 1
// (exact for up to 15 crossings; longer paths are flagged by bit 31 and treated as never equal).
// A room without ceiling polygon (Room._render, miniworld.py:406; the sim-to-real rinks) carries a NEGATIVE height
// word: a ray that crosses its ceiling plane - which every ray that would pass over a wall's top edge does first -
// sees the sky.
template <bool PATH>
__device__ __forceinline__ uint32_t trace_rooms(const float *__restrict__ rooms, int n_rooms, int room, const float *o,
                                                const float *dv, float &t_hit, uint32_t &path) {
    t_hit = INFINITY;
    path = 1;   // leading sentinel bit: the word encodes the length as well as the sides
    if (room < 0) return MAKE_KEY(KIND_SKY, 0, 0);
    const bool xpos = dv[0] > 0, xnz = dv[0] != 0, zpos = dv[2] > 0, znz = dv[2] != 0, ypos = dv[1] > 0, yneg = dv[1] < 0;
    // per-ray reciprocals (one correctly rounded division each); plane distances are (c - o) * inv
    // Sample rays (PATH == false) are part of the bit-exact spec: correctly rounded divisions.  Corner rays
    // (PATH == true) only classify pixels, conservatively by 1/16 pixel, so v_rcp_f32 (1 ulp) is enough.
    const float ix = xnz ? (PATH ? __builtin_amdgcn_rcpf(dv[0]) : 1.0f / dv[0]) : 0.0f;
    const float iy = (ypos || yneg) ? (PATH ? __builtin_amdgcn_rcpf(dv[1]) : 1.0f / dv[1]) : 0.0f;
    const float iz = znz ? (PATH ? __builtin_amdgcn_rcpf(dv[2]) : 1.0f / dv[2]) : 0.0f;
    // a ray parallel to a plane never reaches it: (c - o) * 0 + inf.  (fmaf(a, b, 0) is the rounded product a * b, so
    // the distances are the spec's; written as arithmetic so that the room reads below stay unconditional.)
    const float x_off = xnz ? 0.0f : INFINITY, z_off = znz ? 0.0f : INFINITY;
    const float tfloor = yneg ? (0.0f - o[1]) * iy : INFINITY;
    const int sx = xpos ? 0 : 2, sz = zpos ? 3 : 1;
    // The loop only walks: it has ONE exit and nothing but selects inside; what the ray finally met is worked out
    // after it from the last room's values (a nest of conditionals in the loop body costs dozens of exec-mask
    // excursions per iteration on every lane of the wave).
    float ts = INFINITY, tc = INFINITY, hc = 0.0f, y = 0.0f, height_w = 0.0f, p_lo = 0.0f, p_hi = 0.0f, p_maxy = 0.0f;
    uint32_t nbr = RW_NO_NBR;
    int s = 0, steps = 0;
    for (int iter = 0; iter <= n_rooms; iter++) {
        const float *r = rooms + __umul24((uint32_t)room, MWB_ROOM_WORDS);
        const float4 rect = *(const float4 *)(r + RW_MINX);   // min_x max_x min_z max_z
        height_w = r[RW_HEIGHT];
        const float4 portx = *(const float4 *)(r + RW_SIDE0 + RW_SIDE_WORDS * sx);   // lo hi max_y u_org
        const float4 portz = *(const float4 *)(r + RW_SIDE0 + RW_SIDE_WORDS * sz);
        const uint32_t n01 = (uint32_t)__float_as_int(r[RW_NBR01]), n23 = (uint32_t)__float_as_int(r[RW_NBR23]);
        const uint32_t nbrx = xpos ? (n01 & 0xFFFFu) : (n23 & 0xFFFFu);   // sides 0 / 2
        const uint32_t nbrz = zpos ? (n23 >> 16) : (n01 >> 16);             // sides 3 / 1
        const float tx = fmaf((xpos ? rect.y : rect.x) - o[0], ix, x_off);
        const float tz = fmaf((zpos ? rect.w : rect.z) - o[2], iz, z_off);
        const bool usex = tx <= tz;
        ts = usex ? tx : tz;
        s = usex ? sx : sz;
        tc = ypos ? (fabsf(height_w) - o[1]) * iy : INFINITY;
        p_lo = usex ? portx.x : portz.x; p_hi = usex ? portx.y : portz.y; p_maxy = usex ? portx.z : portz.z;
        nbr = usex ? nbrx : nbrz;
        hc = usex ? fmaf(ts, dv[2], o[2]) : fmaf(ts, dv[0], o[0]);
        y = fmaf(ts, dv[1], o[1]);
        // through the portal of that side (min_y = 0) unless the floor or the ceiling plane comes first
        const bool go_on = nbr != RW_NO_NBR && p_lo < hc && hc < p_hi && 0.0f < y && y < p_maxy && !(tfloor <= ts) && !(tc <= ts) && ts < INFINITY;
        if (!go_on) break;
        if (PATH) { path = (path << 2) | (uint32_t)s; steps++; }
        room = (int)nbr;
    }
    // what stopped the ray, from the last room's values (only floats were carried out of the loop)
    const bool hit_floor = tfloor <= ts, hit_ceil = tc <= ts, escaped = !(ts < INFINITY);
    const bool pass = nbr != RW_NO_NBR && p_lo < hc && hc < p_hi && 0.0f < y && y < p_maxy;
    const bool stopped = hit_floor || hit_ceil || escaped || !pass;   // false: more than n_rooms crossings (nothing is drawn)
    if (PATH && steps > 15) path = 0x80000000u | (uint32_t)steps;
    // same precedence as the sequential tests of the spec: floor, ceiling (sky where the room has none), escape, wall
    const bool wall = stopped && !hit_floor && !hit_ceil && !escaped;
    const bool ceil_drawn = stopped && !hit_floor && hit_ceil && !(height_w < 0.0f);
    const bool floor_ = stopped && hit_floor;
    uint32_t wall_key = MAKE_KEY(KIND_WALL, s, room);
    // a wall with a portal is not convex: tag the convex piece (left / right / above / below the opening) so that
    // the corner-ray classification never spans the opening
    if (PATH) wall_key |= (nbr != RW_NO_NBR ? (hc <= p_lo ? 1u : hc >= p_hi ? 2u : y >= p_maxy ? 3u : 0u) : 0u) << 28;
    t_hit = floor_ ? tfloor : ceil_drawn ? tc : wall ? ts : INFINITY;
    return floor_ ? MAKE_KEY(KIND_FLOOR, 0, room) : ceil_drawn ? MAKE_KEY(KIND_CEIL, 0, room) : wall ? wall_key : MAKE_KEY(KIND_SKY, 0, 0);
}


// The same traversal for rooms that are convex polygons with up to 4 arbitrary edges (YMaze; polygon room table,
// mwb_internal.h).  Spec (polygon rooms): edge k with start p and inward unit normal n is an exit
// candidate if den = fmaf(n.z, d.z, n.x d.x) < 0, at t = fmaf(n.z, p.z - o.z, n.x (p.x - o.x)) / den; the exit edge is the first
// one, in edge order, with the smallest t; the crossing point's distance along it is hc = fmaf(dir.z, hz - p.z, dir.x (hx - p.x))
// with (hx, hz) = fmaf(t, d, o).  Keys and paths carry the edge index where the rectangle version carries the side.
template <bool PATH>
__device__ __forceinline__ uint32_t trace_rooms_poly(const float *__restrict__ rooms, int n_rooms, int room, const float *o,
                                                     const float *dv, float &t_hit, uint32_t &path) {
    t_hit = INFINITY;
    path = 1;
    if (room < 0) return MAKE_KEY(KIND_SKY, 0, 0);
    const bool ypos = dv[1] > 0, yneg = dv[1] < 0;
    const float iy = (ypos || yneg) ? (PATH ? __builtin_amdgcn_rcpf(dv[1]) : 1.0f / dv[1]) : 0.0f;
    const float tfloor = yneg ? (0.0f - o[1]) * iy : INFINITY;
    float ts = INFINITY, tc = INFINITY, hc = 0.0f, y = 0.0f, height_w = 0.0f, p_lo = 0.0f, p_hi = 0.0f, p_maxy = 0.0f;
    int nbr = -1, s = 0, steps = 0;
    for (int iter = 0; iter <= n_rooms; iter++) {
        const float *r = rooms + __umul24((uint32_t)room, MWB_POLY_ROOM_WORDS);
        height_w = r[PW_HEIGHT];
        ts = INFINITY; s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float2 a = *(const float2 *)(r + PW_EDGE0 + PW_EDGE_WORDS * k);       // p.x p.z (the direction is read for the exit edge only)
            const float2 n = *(const float2 *)(r + PW_EDGE0 + PW_EDGE_WORDS * k + 4);   // n.x n.z
            const float den = fmaf(n.y, dv[2], n.x * dv[0]);
            const float num = fmaf(n.y, a.y - o[2], n.x * (a.x - o[0]));
            const float t = PATH ? num * __builtin_amdgcn_rcpf(den) : num / den;
            const bool take = den < 0.0f && t < ts;
            ts = take ? t : ts; s = take ? k : s;
        }
        const float *ed = r + PW_EDGE0 + __umul24((uint32_t)s, PW_EDGE_WORDS);
        const float4 a = *(const float4 *)ed;
        const float4 b = *(const float4 *)(ed + 4);   // n.x n.z lo hi
        p_lo = b.z; p_hi = b.w; p_maxy = ed[8]; nbr = __float_as_int(ed[9]);
        tc = ypos ? (fabsf(height_w) - o[1]) * iy : INFINITY;
        const float hx = fmaf(ts, dv[0], o[0]), hz = fmaf(ts, dv[2], o[2]);
        hc = fmaf(a.w, hz - a.y, a.z * (hx - a.x));
        y = fmaf(ts, dv[1], o[1]);
        const bool go_on = nbr >= 0 && p_lo < hc && hc < p_hi && 0.0f < y && y < p_maxy && !(tfloor <= ts) && !(tc <= ts) && ts < INFINITY;
        if (!go_on) break;
        if (PATH) { path = (path << 2) | (uint32_t)s; steps++; }
        room = nbr;
    }
    const bool hit_floor = tfloor <= ts, hit_ceil = tc <= ts, escaped = !(ts < INFINITY);
    const bool pass = nbr >= 0 && p_lo < hc && hc < p_hi && 0.0f < y && y < p_maxy;
    const bool stopped = hit_floor || hit_ceil || escaped || !pass;
    if (PATH && steps > 15) path = 0x80000000u | (uint32_t)steps;
    const bool wall = stopped && !hit_floor && !hit_ceil && !escaped;
    const bool ceil_drawn = stopped && !hit_floor && hit_ceil && !(height_w < 0.0f);
    const bool floor_ = stopped && hit_floor;
    uint32_t wall_key = MAKE_KEY(KIND_WALL, s, room);
    if (PATH) wall_key |= (nbr >= 0 ? (hc <= p_lo ? 1u : hc >= p_hi ? 2u : y >= p_maxy ? 3u : 0u) : 0u) << 28;
    t_hit = floor_ ? tfloor : ceil_drawn ? tc : wall ? ts : INFINITY;
    return floor_ ? MAKE_KEY(KIND_FLOOR, 0, room) : ceil_drawn ? MAKE_KEY(KIND_CEIL, 0, room) : wall ? wall_key : MAKE_KEY(KIND_SKY, 0, 0);
}

// The room a ray is in after its first n portal crossings (the caller knows that it makes them: rays inside a pixel
// whose corner rays all make them do, see pixel_full).  Same side selection as trace_rooms; nothing is tested.
// Returns -1 if a crossing finds no neighbour (cannot happen for such rays; the caller then starts from the eye's room).
__device__ __forceinline__ int walk_rooms(const float *__restrict__ rooms, int room, const float *o, const float *dv, int n) {
    const bool xpos = dv[0] > 0, xnz = dv[0] != 0, zpos = dv[2] > 0, znz = dv[2] != 0;
    const float ix = xnz ? __builtin_amdgcn_rcpf(dv[0]) : 0.0f, iz = znz ? __builtin_amdgcn_rcpf(dv[2]) : 0.0f;
    const float x_off = xnz ? 0.0f : INFINITY, z_off = znz ? 0.0f : INFINITY;
    for (int i = 0; i < n && room >= 0; i++) {
        const float *r = rooms + __umul24((uint32_t)room, MWB_ROOM_WORDS);
        const float4 rect = *(const float4 *)(r + RW_MINX);
        const uint32_t n01 = (uint32_t)__float_as_int(r[RW_NBR01]), n23 = (uint32_t)__float_as_int(r[RW_NBR23]);
        const float tx = fmaf((xpos ? rect.y : rect.x) - o[0], ix, x_off);
        const float tz = fmaf((zpos ? rect.w : rect.z) - o[2], iz, z_off);
        const uint32_t nbrx = xpos ? (n01 & 0xFFFFu) : (n23 & 0xFFFFu), nbrz = zpos ? (n23 >> 16) : (n01 >> 16);
        const uint32_t nbr = tx <= tz ? nbrx : nbrz;
        room = nbr == RW_NO_NBR ? -1 : (int)nbr;
    }
    return room;
}
__device__ __forceinline__ int walk_rooms_poly(const float *__restrict__ rooms, int room, const float *o, const float *dv, int n) {
    for (int i = 0; i < n && room >= 0; i++) {
        const float *r = rooms + __umul24((uint32_t)room, MWB_POLY_ROOM_WORDS);
        float ts = INFINITY;
        int nbr = -1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float *ed = r + PW_EDGE0 + PW_EDGE_WORDS * k;
            const float den = fmaf(ed[5], dv[2], ed[4] * dv[0]);
            const float t = fmaf(ed[5], ed[1] - o[2], ed[4] * (ed[0] - o[0])) * __builtin_amdgcn_rcpf(den);
            const bool take = den < 0.0f && t < ts;
            ts = take ? t : ts; nbr = take ? __float_as_int(ed[9]) : nbr;
        }
        room = nbr;
    }
    return room;
}

// number of leading portal crossings two path words (trace_rooms<true>) have in common
__device__ __forceinline__ int common_crossings(uint32_t a, uint32_t b) {
    if (((a | b) & 0x80000000u) || a == 0 || b == 0) return 0;   // a path too long for the word / no path (lanes that are no pixel)
    const int ca = __builtin_clz(a), cb = __builtin_clz(b);   // both have their sentinel bit
    const uint32_t x = (a << ca) ^ (b << cb);
    const int eq = x ? __builtin_clz(x) : 32;                  // equal leading bits incl. the sentinel
    const int na = (31 - ca) >> 1, nb = (31 - cb) >> 1, ne = (eq - 1) >> 1;
    const int m = na < nb ? na : nb;
    return ne < m ? ne : m;
}

// the same for four paths at once (a pixel's corner rays): one xor-or-clz over the left-aligned words
__device__ __forceinline__ int common_crossings4(uint32_t a, uint32_t b, uint32_t c, uint32_t e) {
    if (((a | b | c | e) & 0x80000000u) || a == 0 || b == 0 || c == 0 || e == 0) return 0;
    const int ca = __builtin_clz(a), cb = __builtin_clz(b), cc = __builtin_clz(c), ce = __builtin_clz(e);
    const uint32_t A = a << ca;
    const uint32_t x = (A ^ (b << cb)) | (A ^ (c << cc)) | (A ^ (e << ce));
    const int eq = x ? __builtin_clz(x) : 32;
    int cmax = ca > cb ? ca : cb; cmax = cmax > cc ? cmax : cc; cmax = cmax > ce ? cmax : ce;   // the shortest path has the most leading zeros
    const int m = (31 - cmax) >> 1, ne = (eq - 1) >> 1;
    return ne < m ? ne : m;
}

// slab test in box-local axes; returns face 0..5 (-x,+x,-y,+y,-z,+z) or -1
__device__ __forceinline__ int trace_box(const float *fc, const float *dv, float &t_out) {
    const float c = fc[FC_BOX_C], s = fc[FC_BOX_S];
    const float ld[3] = {fmaf(dv[0], c, -(dv[2] * s)), dv[1], fmaf(dv[0], s, dv[2] * c)};
    const float lo_[3] = {fc[FC_BOX_LO], fc[FC_BOX_LO + 1], fc[FC_BOX_LO + 2]};
    const float lo[3] = {-fc[FC_BOX_HX], 0.0f, -fc[FC_BOX_HZ]}, hi[3] = {fc[FC_BOX_HX], fc[FC_BOX_SY], fc[FC_BOX_HZ]};
    float tn = -INFINITY, tf = INFINITY;
    int face = -1;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (ld[a] == 0) { miss = miss || lo_[a] < lo[a] || lo_[a] > hi[a]; continue; }
        float inv = 1.0f / ld[a];
        float t1 = (lo[a] - lo_[a]) * inv, t2 = (hi[a] - lo_[a]) * inv;
        float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
        if (tmin > tn) { tn = tmin; face = a * 2 + (ld[a] > 0 ? 0 : 1); }
        if (tmax < tf) tf = tmax;
    }
    if (miss || face < 0 || !(tn <= tf) || !(tn > 0)) return -1;
    t_out = tn;
    return face;
}

// ---- mesh entities and frames (entity tasks; DESIGN.md 5, "meshes" / "frames": the written arithmetic) -------------------------
#ifndef ENT_WGS_PER_CU
#define ENT_WGS_PER_CU 4   // the entity kernels' residency target: 128 VGPRs (5 workgroups at 96 VGPRs spill 656 B/lane and run no faster)
#endif
#define KIND_MESH 5u
#define KIND_FRAME 6u
__device__ __forceinline__ float dot3f(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ void cross3f(const float *a, const float *b, float *o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
// the ray direction in the mesh's frame: R^T d / scale (the ray parameter stays the world's)
__device__ __forceinline__ void mesh_local_dir(const float *blk, const float *dv, float *ld) {
    const float c = blk[FC_BOX_C], s = blk[FC_BOX_S], inv_s = blk[FC_LIT_BOX + FE_MESH_INVS];
    ld[0] = fmaf(dv[0], c, -(dv[2] * s)) * inv_s; ld[1] = dv[1] * inv_s; ld[2] = fmaf(dv[0], s, dv[2] * c) * inv_s;
}
// Moeller-Trumbore against one triangle record (v0, e1, e2): u, v (not divided), det; front faces only (det > 0)
__device__ __forceinline__ bool mesh_tri(const float *lo, const float *ld, const float4 r0, const float4 r1, const float4 r2, bool need_inside,
                                         float &t, float &u, float &v, float &det) {
    const float v0[3] = {r0.x, r0.y, r0.z}, e1[3] = {r0.w, r1.x, r1.y}, e2[3] = {r1.z, r1.w, r2.x};
    float pv[3], qv[3];
    cross3f(ld, e2, pv);
    det = dot3f(e1, pv);
    if (!(det > 0.0f)) return false;
    const float tv[3] = {lo[0] - v0[0], lo[1] - v0[1], lo[2] - v0[2]};
    u = dot3f(tv, pv);
    cross3f(tv, e1, qv);
    v = dot3f(ld, qv);
    if (need_inside && (u < 0.0f || u > det || v < 0.0f || u + v > det)) return false;
    t = dot3f(e2, qv) / det;
    return need_inside ? t > 0.0f : true;
}
// nearest front-facing triangle of the entity's mesh along (eye, dv) that is strictly nearer than t_max: its index in draw order
// or -1.  Threaded BVH (host-built, gym_miniworld_amd/meshes.py): depth-first node order, `skip` links, no stack.
#define MQ_CAP 320     // mesh-pixel queue entries per wave: batches wait for the end of the frame, where all waves share them
#define MB_HALF 4      // pixels_mesh: samples per round
#define MB_TASKS 320   //   (ray, mesh) pairs per round
#define MB_WAVE_BYTES (MB_HALF * 64 * 8 + MB_TASKS * 2 + 64 * 4 + 16)   // per wave: slots, pairs, pixel coordinates, counter
typedef float f4n __attribute__((ext_vector_type(4)));   // a plain 16-byte vector for the mesh records (walk_meshes)
// ImageFrame / TextFrame: slab [0, depth] x [-h/2, h/2] x [-w/2, w/2] in the frame's axes; returns the character cell of the
// front (+x) face, 100 for a black side, -1 for a miss or the missing back
__device__ __forceinline__ int trace_frame(const float *blk, const float *dv, float &t_out) {
    const float c = blk[FC_BOX_C], s = blk[FC_BOX_S];
    const float *fe = blk + FC_LIT_BOX;
    const float ld[3] = {fmaf(dv[0], c, -(dv[2] * s)), dv[1], fmaf(dv[0], s, dv[2] * c)};
    const float lo_[3] = {blk[FC_BOX_LO], blk[FC_BOX_LO + 1], blk[FC_BOX_LO + 2]};
    const float lo[3] = {0.0f, -fe[FE_FRAME_HY], -fe[FE_FRAME_HZ]}, hi[3] = {fe[FE_FRAME_SX], fe[FE_FRAME_HY], fe[FE_FRAME_HZ]};
    float tn = -INFINITY, tf = INFINITY;
    int face = -1;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (ld[a] == 0) { miss = miss || lo_[a] < lo[a] || lo_[a] > hi[a]; continue; }
        const float inv = 1.0f / ld[a];
        const float t1 = (lo[a] - lo_[a]) * inv, t2 = (hi[a] - lo_[a]) * inv;
        const float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
        if (tmin > tn) { tn = tmin; face = a * 2 + (ld[a] > 0 ? 0 : 1); }
        if (tmax < tf) tf = tmax;
    }
    if (miss || face <= 0 || !(tn <= tf) || !(tn > 0)) return -1;   // face 0 = the -x side: not drawn
    t_out = tn;
    if (face != 1) return 100;
    const float z = fmaf(tn, ld[2], lo_[2]);
    const int n_ch = __float_as_int(fe[FE_FRAME_NCH]);
    int i = (int)floorf((fe[FE_FRAME_HZ] - z) / fe[FE_FRAME_CW]);
    return i < 0 ? 0 : (i >= n_ch ? n_ch - 1 : i);
}
__device__ __forceinline__ bool frame_front_tc(const float *blk, int cell, const float *dv, float &s_, float &t_) {
    const float c = blk[FC_BOX_C], s = blk[FC_BOX_S];
    const float *fe = blk + FC_LIT_BOX;
    const float ld[3] = {fmaf(dv[0], c, -(dv[2] * s)), dv[1], fmaf(dv[0], s, dv[2] * c)};
    const float lo_[3] = {blk[FC_BOX_LO], blk[FC_BOX_LO + 1], blk[FC_BOX_LO + 2]};
    const float tt = (fe[FE_FRAME_SX] - lo_[0]) / ld[0];
    const float y = fmaf(tt, ld[1], lo_[1]), z = fmaf(tt, ld[2], lo_[2]);
    const float z1 = fe[FE_FRAME_HZ] - fe[FE_FRAME_CW] * (float)cell;
    s_ = (z1 - z) / fe[FE_FRAME_CW]; t_ = (y + fe[FE_FRAME_HY]) / (2.0f * fe[FE_FRAME_HY]);
    return ld[0] != 0.0f && tt > 0.0f;
}

// ---- shading: continuous in its inputs, so fused / approximate arithmetic is within the +-1 LSB bar
struct TexLds { int w, h, n_levels; float sc_s, sc_t; int pad[3]; uint32_t off[MWB_MAX_LEVELS]; };
static_assert(sizeof(TexLds) == sizeof(MwbTexDesc), "TexLds mirrors MwbTexDesc");

// No "fp contract(fast)" here: which products the compiler fuses would depend on the code around each inlined copy, and
// the interior-pixel path must produce the very bits the 8-sample path produces for the same surface
// (test_fast_path_equals_full_sample_path).  Every fused multiply-add below is written out.
__device__ __forceinline__ void bilinear(const uint32_t *__restrict__ texels, uint32_t off, int w, int h, float s, float t, float *rgb) {
    float uu = fmaf(s, (float)w, -0.5f), vv = fmaf(t, (float)h, -0.5f);
    float fu = floorf(uu), fv = floorf(vv);
    float a = uu - fu, b = vv - fv;
    int i0 = (int)fu, j0 = (int)fv;
    // s,t in [0,1): i0 in [-1, w-1]; wrap with compares instead of integer division
    int i1 = i0 + 1; if (i1 >= w) i1 -= w; if (i0 < 0) i0 += w;
    int j1 = j0 + 1; if (j1 >= h) j1 -= h; if (j0 < 0) j0 += h;
    // unsigned 32-bit texel indices (24-bit multiplies: levels are at most 1024 wide) keep the address math
    // out of 64-bit VALU arithmetic: the loads use the scalar base + 32-bit offset form
    // (the pyramids total ~14 MB, so byte offsets fit 32 bits)
    const uint32_t r0 = off + __umul24((uint32_t)j0, (uint32_t)w), r1 = off + __umul24((uint32_t)j1, (uint32_t)w);
    const char *tb = (const char *)texels;
    auto texel = [tb](uint32_t idx) { return *(const uint32_t *)(tb + (size_t)(idx << 2)); };
    uint32_t t00 = texel(r0 + (uint32_t)i0), t10 = texel(r0 + (uint32_t)i1), t01 = texel(r1 + (uint32_t)i0), t11 = texel(r1 + (uint32_t)i1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float c00 = (float)((t00 >> (8 * k)) & 255u), c10 = (float)((t10 >> (8 * k)) & 255u);
        float c01 = (float)((t01 >> (8 * k)) & 255u), c11 = (float)((t11 >> (8 * k)) & 255u);
        rgb[k] = fmaf(c11, w11, fmaf(c01, w01, fmaf(c10, w10, c00 * w00)));
    }
}

__device__ __forceinline__ void sample_texture(const uint32_t *__restrict__ texels, const TexLds &T, float s, float t, float sx,
                                               float tx, float sy, float ty, bool valid, float *rgb) {
    // straight-line code (selects, no divergent branches): the shading batches run with dense lanes and every
    // exec-mask excursion costs issue slots on all of them
    const int4 dims = *(const int4 *)&T.w;   // w h n_levels sc_s(bits)
    const int maxl = dims.z - 1;
    const float fw = (float)dims.x, fh = (float)dims.y;
    const float dsdx = (sx - s) * fw, dtdx = (tx - t) * fh;
    const float dsdy = (sy - s) * fw, dtdy = (ty - t) * fh;
    const float r1 = fmaf(dsdx, dsdx, dtdx * dtdx), r2 = fmaf(dsdy, dsdy, dtdy * dtdy);
    const float rho2 = r1 > r2 ? r1 : r2;
    const float lg = 0.5f * __log2f(fmaxf(rho2, 1.0f));          // rho2 <= 1 (magnification): level 0; NaN -> 1 -> 0
    const float lambda = (valid && rho2 < INFINITY) ? lg : (float)maxl;
    float ws = s - floorf(s), wt = t - floorf(t);   // GL_REPEAT
    // non-finite coordinates (a plane seen exactly edge-on) must not become texel addresses
    ws = (ws >= 0.0f && ws < 1.0f) ? ws : 0.0f;
    wt = (wt >= 0.0f && wt < 1.0f) ? wt : 0.0f;
    const float fl = floorf(lambda);
    int l0 = (int)fl; l0 = l0 > maxl ? maxl : l0;
    const int l1 = l0 + 1 > maxl ? maxl : l0 + 1;
    const float fr = (l0 == maxl) ? 0.0f : lambda - fl;
    // both levels are always fetched (l1 == l0 when there is nothing to blend) so that the eight texel
    // loads are issued together and waited for once
    float c0[3], c1[3];
    int w0 = dims.x >> l0; w0 = w0 < 1 ? 1 : w0;
    int h0 = dims.y >> l0; h0 = h0 < 1 ? 1 : h0;
    int w1 = dims.x >> l1; w1 = w1 < 1 ? 1 : w1;
    int h1 = dims.y >> l1; h1 = h1 < 1 ? 1 : h1;
    bilinear(texels, T.off[l0], w0, h0, ws, wt, c0);
    bilinear(texels, T.off[l1], w1, h1, ws, wt, c1);
#pragma unroll
    for (int k = 0; k < 3; k++) rgb[k] = fmaf(c1[k] - c0[k], fr, c0[k]);
}

// texture coordinates where ray (o, dv) meets the plane {axis = plane}; false if behind / parallel
__device__ __forceinline__ bool plane_texcoord(int axis, float plane, float u_org, float u_scale, float sc_t, const float *o,
                                               const float *dv, float &s, float &t) {
    float od = axis == 0 ? dv[0] : axis == 1 ? dv[1] : dv[2];
    float oo = axis == 0 ? o[0] : axis == 1 ? o[1] : o[2];
    float tt = (plane - oo) * __builtin_amdgcn_rcpf(od);
    float px = fmaf(tt, dv[0], o[0]), py = fmaf(tt, dv[1], o[1]), pz = fmaf(tt, dv[2], o[2]);
    // floor / ceiling: (x, z) * scale; walls: (distance along the edge, y) * scale (miniworld.py:19-68)
    float a = axis == 0 ? pz : px;
    float b = axis == 1 ? pz : py;
    s = (a - u_org) * u_scale; t = b * sc_t;
    return od != 0.0f && tt > 0.0f;
}

template <int NBOX, bool POLY>
struct RenderCtx {
    const float *rooms, *fc;
    const TexLds *tex;
    const uint32_t *texels;
    uint8_t *fb;
    float *depth;   // this env's depth map or null
    Cam cam;
    int n_rooms, cam_room, W, H, layout;
    int depth_stride;   // floats between rows of the depth map: W, or the whole view's width when this workgroup renders a tile of it
    float cull_cc[1], cull_oc[1][3], zA, zB;   // the box's cull constants (NBOX == 1 only: scalar registers)
    bool boxes_in_view;   // workgroup-uniform (scalar): false = no ray of this frame can touch a box
    const MwbMeshDesc *mesh_desc;   // entity tasks: mesh geometries in HBM (L2 resident), or null
    const uint4 *mdesc;             // the walk's part of every descriptor, staged in LDS: node_off, tri_off, n_nodes, n_orders
    // this wave's scratch for a batch of mesh pixels (pixels_mesh): per (sample-in-half, lane) the nearest surface so far as an
    // ordered 64-bit word, the (ray, mesh) pairs that passed their gate, the batch's pixel coordinates, the pair counter
    unsigned long long *mb_slots;   // [MB_HALF][64]
    uint16_t *mb_tasks;             // [MB_TASKS]
    uint32_t *mb_pix;               // [64]
    int *mb_count;
    const float4 *mesh_data;
    int exp_flags;                  // MWB_EXP experiment switches (timing experiments only)
    uint32_t mesh_slots;            // bit b: entity slot b is a mesh in this frame (workgroup-uniform)
    unsigned long long *dbg_counters;   // MWB_EXP bit 2: [0] sample rays in walk_meshes, [1] walks, [2] node visits, [3] triangle tests, [4] wave loop iterations, [5] wave calls
    // per work item (15 x 15 pixels), which boxes can touch any of its rays at all (frame-level pre-test pass; null = unknown)
    const uint4 *item_res;
    int part_h_inv;       // ceil(65536 / rows per item): row -> quarter without a division
    __device__ __forceinline__ uint32_t item_boxes(int px, int py) const {
        if (NBOX == 1 || !item_res) return 0xFFFFFFFFu;   // one box: the frame-level gate and the per-ray sphere test are enough
        const int item = ((px * 4370) >> 16) * 4 + ((py * part_h_inv) >> 16);   // px / 15 (exact below 4755), py / part_h
        return item_res[item].z;
    }

    // rays through the +1 pixel neighbours (for the LOD differences): the ray is affine in the window
    // coordinates, so they are the centre ray plus a per-frame constant (shading-only, tolerance-bound)
    __device__ __forceinline__ static void neighbour_rays_of(const Cam &cam, const float *dc, float *dx, float *dy) {
        const float sx = 2.0f * cam.invW * cam.TW, sy = 2.0f * cam.invH * cam.TH;
#pragma unroll
        for (int k = 0; k < 3; k++) { dx[k] = fmaf(cam.S[k], sx, dc[k]); dy[k] = fmaf(cam.U[k], sy, dc[k]); }
    }
    __device__ __forceinline__ void neighbour_rays(const float *dc, float *dx, float *dy) const {
        const float sx = 2.0f * cam.invW * cam.TW, sy = 2.0f * cam.invH * cam.TH;
#pragma unroll
        for (int k = 0; k < 3; k++) { dx[k] = fmaf(cam.S[k], sx, dc[k]); dy[k] = fmaf(cam.U[k], sy, dc[k]); }
    }

    template <bool PATH>
    __device__ __forceinline__ uint32_t trace_from(int room, const float *dv, float &t_hit, uint32_t &path) const {
        if constexpr (POLY) return trace_rooms_poly<PATH>(rooms, n_rooms, room, cam.eye, dv, t_hit, path);
        else return trace_rooms<PATH>(rooms, n_rooms, room, cam.eye, dv, t_hit, path);
    }
    template <bool PATH>
    __device__ __forceinline__ uint32_t trace(const float *dv, float &t_hit, uint32_t &path) const {
        return trace_from<PATH>(cam_room, dv, t_hit, path);
    }

    // shade() for the polygon room table: a wall's plane, texture origin and lit colour come from its edge record
    template <bool INTERIOR>
    __device__ __forceinline__ void shade_poly(uint32_t key, int first_k, float cx, float cy, float *col) const {
        const uint32_t kind = key & 7u, side = (key >> 3) & 3u;
        if (!INTERIOR) {
            if (kind == KIND_SKY) { col[0] = fc[FC_SKY]; col[1] = fc[FC_SKY + 1]; col[2] = fc[FC_SKY + 2]; return; }
            if (kind == KIND_BOX) {
                const float *lb = fc + FC_LIT_BOX + 3 * ((key >> 3) & 7u) + (NBOX > 1 ? (key >> 6) * FC_BOX_STRIDE : 0u);
                col[0] = lb[0]; col[1] = lb[1]; col[2] = lb[2];
                return;
            }
        }
        const float *r = rooms + __umul24(key >> 6, MWB_POLY_ROOM_WORDS);
        const float height = fabsf(r[PW_HEIGHT]);
        const uint32_t texw = (uint32_t)__float_as_int(r[PW_TEX]);
        const float *ed = r + PW_EDGE0 + __umul24(side, PW_EDGE_WORDS);
        const float4 a = *(const float4 *)ed;          // p.x p.z dir.x dir.z
        const float2 n = *(const float2 *)(ed + 4);    // n.x n.z
        const bool wall = kind == KIND_WALL, floor_ = kind == KIND_FLOOR;
        const uint32_t tex_id = (wall ? texw : floor_ ? (texw >> 8) : (texw >> 16)) & 255u;
        const TexLds &T = tex[tex_id];
        const float sc_s = T.sc_s, sc_t = T.sc_t;
        const float plane = floor_ ? 0.0f : height;
        auto texcoord = [&](const float *dv, float &s, float &t) {
            if (wall) {
                const float den = fmaf(n.y, dv[2], n.x * dv[0]);
                const float tt = fmaf(n.y, a.y - cam.eye[2], n.x * (a.x - cam.eye[0])) * __builtin_amdgcn_rcpf(den);
                const float hx = fmaf(tt, dv[0], cam.eye[0]), hz = fmaf(tt, dv[2], cam.eye[2]), yy = fmaf(tt, dv[1], cam.eye[1]);
                s = fmaf(a.w, hz - a.y, a.z * (hx - a.x)) * sc_s; t = yy * sc_t;
                return den != 0.0f && tt > 0.0f;
            }
            return plane_texcoord(1, plane, 0.0f, sc_s, sc_t, cam.eye, dv, s, t);
        };
        float dc[3], dx[3], dy[3];
        make_ray(cam, cx, cy, dc);
        neighbour_rays(dc, dx, dy);
        float s0, t0, s1, t1, s2, t2;
        const bool hit0 = texcoord(dc, s0, t0);
        const bool v1 = texcoord(dx, s1, t1);
        const bool v2 = texcoord(dy, s2, t2);
        bool valid = v1 && v2;
        if (!INTERIOR && !hit0) {
            float dv[3];
            make_ray(cam, cx + c_sample_x[first_k], cy + c_sample_y[first_k], dv);
            texcoord(dv, s0, t0);
            valid = false; s1 = s2 = s0; t1 = t2 = t0;
        }
        float texel[3];
        sample_texture(texels, T, s0, t0, s1, t1, s2, t2, valid, texel);
        {   // flat face with the edge's inward normal (fixed-function lighting, miniworld.py:1026-1045); evaluated after the
            // texel fetch, the normal read again from LDS, so that no lit colour is carried through the fetch (registers)
            const float2 nn = *(const float2 *)(ed + 4);
            const float *Ld = fc + FC_LIGHT_DIR, *amb = fc + FC_LIGHT_AMB, *dif = fc + FC_LIGHT_DIF;
            float ndl = nn.x * Ld[0] + 0.0f * Ld[1] + nn.y * Ld[2];
            ndl = ndl < 0.0f ? 0.0f : ndl;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float v = (0.2f + amb[q]) + ndl * dif[q];
                const float wv = v > 1.0f ? 1.0f : v;
                const float lit = wall ? wv : (floor_ ? fc[FC_LIT_FLOOR + q] : fc[FC_LIT_CEIL + q]);
                col[q] = lit * (texel[q] * (1.0f / 255.0f));
            }
        }
    }

    // INTERIOR: the caller guarantees a room surface (floor / ceiling / wall) whose plane the pixel's centre ray meets
    // in front of the eye (the four corner rays of the pixel do): no sky / box cases, no fallback to a sample's ray.
    // Every mesh entity whose gate sphere the ray (eye, dv) passes, walked by this lane on its own: ONE loop in which a lane either
    // visits the next node of the mesh it is in or moves on to its next candidate - so a wave's time is its slowest lane's own
    // work, not the sum over all the meshes any of its 64 pixels touches (a wave-uniform loop over the entities cost 5-15x more:
    // scripts/ab_mesh_phases.py).  cand: the lane's candidate slots; th / key: the nearest surface so far, updated.  Slot order
    // among meshes is kept (a later mesh wins only when strictly nearer; within a mesh the first triangle drawn wins a tie).
    __device__ __forceinline__ void walk_meshes(uint32_t cand, const float *dv, float dd, float &th, uint32_t &key) const {
        const f4n *gd = (const f4n *)mesh_data;
        const f4n *nodes = gd, *tris = gd;
        float lo[3] = {0, 0, 0}, ld[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
        int node = 0, n_nodes = 0, bi = 0, best = -1;
        int leaf_first = 0, leaf_cnt = 0;   // a leaf this lane has reached and not yet tested
        unsigned c_walks = 0, c_visits = 0, c_tris = 0, c_iters = 0;   // MWB_EXP bit 2: counters (timing experiments only)
        for (;;) {
            // (a) the lanes with no mesh in progress take their next candidate whose gate the ray passes.  Kept OUT of the node loop: in
            // one merged loop every iteration of the wave paid for this path too - ~100 instructions and a descriptor fetch - because
            // some lane or other is always between two meshes.
            while (node >= n_nodes && cand) {
                bi = __builtin_ctz(cand);
                cand &= cand - 1u;
                best = -1;
                const float *blk = fc + bi * FC_BOX_STRIDE;
                const float cc = blk[FC_CULL_CC];
                const float b = dv[0] * blk[FC_CULL_OC] + dv[1] * blk[FC_CULL_OC + 1] + dv[2] * blk[FC_CULL_OC + 2];
                node = n_nodes = 0;
                if (!(cc <= 0.0f || (b > 0.0f && b * b >= dd * cc))) continue;
                mesh_local_dir(blk, dv, ld);
                lo[0] = blk[FC_BOX_LO]; lo[1] = blk[FC_BOX_LO + 1]; lo[2] = blk[FC_BOX_LO + 2];
                // the node boxes are padded by 1e-4 of the mesh (host): a 1-ulp reciprocal cannot lose a real hit
                inv[0] = __builtin_amdgcn_rcpf(ld[0]); inv[1] = __builtin_amdgcn_rcpf(ld[1]); inv[2] = __builtin_amdgcn_rcpf(ld[2]);
                const uint4 md = mdesc[__float_as_int(blk[FC_LIT_BOX + FE_MESH_GEOM])];   // node_off, tri_off, n_nodes, n_orders (LDS)
                // the threading that visits the nearer child first for this ray's direction (bit a: component a negative)
                const uint32_t oct = md.w == 8u ? ((ld[0] < 0.0f ? 1u : 0u) | (ld[1] < 0.0f ? 2u : 0u) | (ld[2] < 0.0f ? 4u : 0u)) : 0u;
                nodes = gd + md.x + (size_t)(oct * 2u * md.z); tris = gd + md.y; n_nodes = (int)md.z;
                c_walks++;
            }
            if (!__any(node < n_nodes)) break;
            // (b) node steps only - the loop body a wave repeats most often stays short - until every lane stands at a leaf or at the
            // end of its mesh.  The records of a node AND of the one behind it (its first child, if it has children: depth-first
            // order) are fetched together: a descent - the commonest move - costs no second round trip to L2.
            while (leaf_cnt == 0 && node < n_nodes) {
                c_iters++;
                const int nx = node + 1 < n_nodes ? node + 1 : node;
                const f4n a = nodes[2 * node], bb = nodes[2 * node + 1], a2 = nodes[2 * nx], b2 = nodes[2 * nx + 1];
                auto visit = [&](const f4n &a, const f4n &bb) {
                    c_visits++;
                    float t0 = (a.x - lo[0]) * inv[0], t1 = (bb.x - lo[0]) * inv[0];
                    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                    t0 = (a.y - lo[1]) * inv[1]; t1 = (bb.y - lo[1]) * inv[1];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    t0 = (a.z - lo[2]) * inv[2]; t1 = (bb.z - lo[2]) * inv[2];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    const uint32_t fcnt = (uint32_t)__float_as_int(bb.w);
                    const bool hit = tn <= tf * 1.00001f && tf > 0.0f && tn <= th;
                    const int cnt = (int)(fcnt >> 24);
                    const bool down = hit && cnt == 0;
                    node = down ? node + 1 : __float_as_int(a.w);   // into the first child, or on past this subtree / leaf
                    if (hit && cnt) { leaf_first = (int)(fcnt & 0xFFFFFFu); leaf_cnt = cnt; }
                    return down;
                };
                if (visit(a, bb)) visit(a2, b2);
            }
            // (c) the lanes that stand at a leaf test its triangles (spec arithmetic: correctly rounded t).  The record of
            // triangle q + 1 is requested before triangle q is tested: one exposed round trip per leaf, not one per triangle.
            if (leaf_cnt) {
                c_iters++;
                f4n p0 = tris[3 * leaf_first], p1 = tris[3 * leaf_first + 1], p2 = tris[3 * leaf_first + 2];
                for (int q = 0; q < leaf_cnt; q++) {
                    const f4n q0 = p0, q1 = p1, q2 = p2;
                    if (q + 1 < leaf_cnt) { p0 = tris[3 * (leaf_first + q + 1)]; p1 = tris[3 * (leaf_first + q + 1) + 1]; p2 = tris[3 * (leaf_first + q + 1) + 2]; }
                    const float4 r0 = make_float4(q0.x, q0.y, q0.z, q0.w), r1 = make_float4(q1.x, q1.y, q1.z, q1.w), r2 = make_float4(q2.x, q2.y, q2.z, q2.w);
                    float t, u, v, det;
                    c_tris++;
                    if (mesh_tri(lo, ld, r0, r1, r2, true, t, u, v, det)) {
                        const int idx = __float_as_int(r2.y);
                        if (t < th || (t == th && best >= 0 && idx < best)) {
                            th = t; best = idx;
                            key = KIND_MESH | ((uint32_t)bi << 3) | ((uint32_t)idx << 8);
                        }
                    }
                }
            }
            leaf_cnt = 0;
        }
        if ((exp_flags & 4) && dbg_counters) {
            atomicAdd(dbg_counters + 0, 1ull); atomicAdd(dbg_counters + 1, (unsigned long long)c_walks); atomicAdd(dbg_counters + 2, (unsigned long long)c_visits);
            atomicAdd(dbg_counters + 3, (unsigned long long)c_tris);
            unsigned mx = c_iters;   // the wave's loop count = its slowest lane's
            for (int o = 32; o; o >>= 1) { const unsigned v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
            if ((threadIdx.x & 63) == 0) { atomicAdd(dbg_counters + 4, (unsigned long long)mx); atomicAdd(dbg_counters + 5, 1ull); }
        }
    }

    // One shade per (pixel, triangle) with the attributes evaluated at the PIXEL CENTRE (extrapolated, as a multisampling
    // rasteriser without centroid sampling does): per-vertex fixed-function lighting with the normal R n / scale (not
    // renormalised: GL_NORMALIZE is off), clamped to 1, interpolated with the centre ray's barycentrics; where the centre ray sees
    // the triangle's back or edge, the covering sample's ray instead; a textured mesh modulates by its image.
    // (static, every input an argument: a non-inlined MEMBER function would take `this`, and a context whose address escapes lives in
    // scratch memory for the whole kernel - 40 % slower when tried on another function)
    __device__ __noinline__ static void shade_mesh_s(const float *fc, const MwbMeshDesc *mesh_desc, const float4 *mesh_data, const TexLds *tex, const uint32_t *texels,
                                                     const Cam *cam_p, int exp_flags, uint32_t key, int first_k, float cx, float cy, float *col) {
        const Cam &cam = *cam_p;
        if (exp_flags & 2) { col[0] = 0.5f; col[1] = 0.5f; col[2] = 0.5f; return; }   // experiment: flat grey meshes
        const int bi = (int)((key >> 3) & 31u), tri = (int)(key >> 8);
        const float *blk = fc + bi * FC_BOX_STRIDE;
        const float *me = blk + FC_LIT_BOX;
        const MwbMeshDesc &md = mesh_desc[__float_as_int(me[FE_MESH_GEOM])];
        const float4 *rec = mesh_data + md.tri2_off + 3 * tri, *sh = mesh_data + md.shade_off + 4 * tri;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3];
        const float nrm[3][3] = {{s0.x, s0.y, s0.z}, {s0.w, s1.x, s1.y}, {s1.z, s1.w, s2.x}};
        const float tc[6] = {s2.y, s2.z, s2.w, s3.x, s3.y, s3.z};
        float vc[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float ndl = nrm[k][0] * me[FE_MESH_LL] + nrm[k][1] * me[FE_MESH_LL + 1] + nrm[k][2] * me[FE_MESH_LL + 2];
            ndl = ndl < 0.0f ? 0.0f : ndl;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float v = me[FE_MESH_AMB + q] + ndl * me[FE_MESH_DIF + q] * me[FE_MESH_KD + q];
                vc[k][q] = v > 1.0f ? 1.0f : v;
            }
        }
        const float lo[3] = {blk[FC_BOX_LO], blk[FC_BOX_LO + 1], blk[FC_BOX_LO + 2]};
        float dc[3], ld[3], t, u, v, det;
        make_ray(cam, cx, cy, dc);
        mesh_local_dir(blk, dc, ld);
        const bool centre_ok = mesh_tri(lo, ld, r0, r1, r2, false, t, u, v, det);
        if (!centre_ok) {
            float ds[3];
            make_ray(cam, cx + c_sample_x[first_k], cy + c_sample_y[first_k], ds);
            mesh_local_dir(blk, ds, ld);
            mesh_tri(lo, ld, r0, r1, r2, false, t, u, v, det);
        }
        const float ub = u / det, vb = v / det;
#pragma unroll
        for (int q = 0; q < 3; q++) col[q] = fmaf(vb, vc[2][q] - vc[0][q], fmaf(ub, vc[1][q] - vc[0][q], vc[0][q]));
        const int tex_id = __float_as_int(me[FE_MESH_TEX]);
        if (tex_id >= 0) {
            const float s_0 = fmaf(vb, tc[4] - tc[0], fmaf(ub, tc[2] - tc[0], tc[0])), t_0 = fmaf(vb, tc[5] - tc[1], fmaf(ub, tc[3] - tc[1], tc[1]));
            float s_1 = s_0, t_1 = t_0, s_2 = s_0, t_2 = t_0;
            bool valid = false;
            if (centre_ok) {
                float dx[3], dy[3], l1[3], l2[3], tt, u1, v1, d1, u2, v2, d2;
                neighbour_rays_of(cam, dc, dx, dy);
                mesh_local_dir(blk, dx, l1); mesh_local_dir(blk, dy, l2);
                const bool ok1 = mesh_tri(lo, l1, r0, r1, r2, false, tt, u1, v1, d1), ok2 = mesh_tri(lo, l2, r0, r1, r2, false, tt, u2, v2, d2);
                valid = ok1 && ok2;
                if (valid) {
                    u1 /= d1; v1 /= d1; u2 /= d2; v2 /= d2;
                    s_1 = fmaf(v1, tc[4] - tc[0], fmaf(u1, tc[2] - tc[0], tc[0])); t_1 = fmaf(v1, tc[5] - tc[1], fmaf(u1, tc[3] - tc[1], tc[1]));
                    s_2 = fmaf(v2, tc[4] - tc[0], fmaf(u2, tc[2] - tc[0], tc[0])); t_2 = fmaf(v2, tc[5] - tc[1], fmaf(u2, tc[3] - tc[1], tc[1]));
                }
            }
            float texel[3];
            sample_texture(texels, tex[tex_id], s_0, t_0, s_1, t_1, s_2, t_2, valid, texel);
#pragma unroll
            for (int q = 0; q < 3; q++) col[q] = col[q] * (texel[q] * (1.0f / 255.0f));
        }
    }
    // the front of an ImageFrame / TextFrame (its picture, or one texture per character; a blank cell is plain lit white) or a black side
    __device__ __noinline__ static void shade_frame_s(const float *fc, const TexLds *tex, const uint32_t *texels, const Cam *cam_p, uint32_t key, int first_k, float cx, float cy, float *col) {
        const Cam &cam = *cam_p;
        const int bi = (int)((key >> 3) & 31u), code = (int)(key >> 8);
        const float *blk = fc + bi * FC_BOX_STRIDE;
        const float *fe = blk + FC_LIT_BOX;
        if (code >= 100) { col[0] = 0.0f; col[1] = 0.0f; col[2] = 0.0f; return; }
        const int tex_id = __float_as_int(fe[FE_FRAME_TEX + code]);
        if (tex_id < 0) { col[0] = fe[FE_FRAME_LIT]; col[1] = fe[FE_FRAME_LIT + 1]; col[2] = fe[FE_FRAME_LIT + 2]; return; }
        float dc[3], dx[3], dy[3], s0, t0, s1, t1, s2, t2;
        make_ray(cam, cx, cy, dc);
        neighbour_rays_of(cam, dc, dx, dy);
        bool valid;
        if (!frame_front_tc(blk, code, dc, s0, t0)) {
            float ds[3];
            make_ray(cam, cx + c_sample_x[first_k], cy + c_sample_y[first_k], ds);
            frame_front_tc(blk, code, ds, s0, t0);
            valid = false; s1 = s2 = s0; t1 = t2 = t0;
        } else {
            const bool v1 = frame_front_tc(blk, code, dx, s1, t1), v2 = frame_front_tc(blk, code, dy, s2, t2);
            valid = v1 && v2;
        }
        float texel[3];
        sample_texture(texels, tex[tex_id], s0, t0, s1, t1, s2, t2, valid, texel);
#pragma unroll
        for (int q = 0; q < 3; q++) col[q] = fe[FE_FRAME_LIT + q] * (texel[q] * (1.0f / 255.0f));
    }

    template <bool INTERIOR>
    __device__ __forceinline__ void shade(uint32_t key, int first_k, float cx, float cy, float *col) const {
        if constexpr (POLY) { shade_poly<INTERIOR>(key, first_k, cx, cy, col); return; }
        const uint32_t kind = key & 7u, side = (key >> 3) & 3u;
        if constexpr (!INTERIOR && NBOX > MWB_MAX_BOXES) {   // entity tasks
            if (kind == KIND_MESH) {
                if (exp_flags & 8) { col[0] = col[1] = col[2] = 0.5f; return; }   // experiment: no call at all (the cost of the call itself)
                const Cam cam_copy = cam;   // only this copy's address leaves the function
                shade_mesh_s(fc, mesh_desc, mesh_data, tex, texels, &cam_copy, exp_flags, key, first_k, cx, cy, col);
                return;
            }
            if (kind == KIND_FRAME) { const Cam cam_copy = cam; shade_frame_s(fc, tex, texels, &cam_copy, key, first_k, cx, cy, col); return; }
        }
        if (!INTERIOR) {
            if (kind == KIND_SKY) { col[0] = fc[FC_SKY]; col[1] = fc[FC_SKY + 1]; col[2] = fc[FC_SKY + 2]; return; }
            if (kind == KIND_BOX) {   // the key's room field holds the box index
                const float *lb = fc + FC_LIT_BOX + 3 * ((key >> 3) & 7u) + (NBOX > 1 ? (key >> 6) * FC_BOX_STRIDE : 0u);
                col[0] = lb[0]; col[1] = lb[1]; col[2] = lb[2];
                return;
            }
        }
        // every read below is unconditional (any side index / texture slot is a valid address) and the choices are
        // selects: the batch runs with dense lanes, divergent excursions would be paid by all of them
        const float *r = rooms + __umul24(key >> 6, MWB_ROOM_WORDS);
        const float4 rect = *(const float4 *)(r + RW_MINX);   // min_x max_x min_z max_z
        const float height = r[RW_HEIGHT];
        const uint32_t texw = (uint32_t)__float_as_int(r[RW_TEX]);
        const float u_org_w = r[RW_SIDE0 + RW_SIDE_WORDS * side + RS_UORG];
        const bool wall = kind == KIND_WALL, floor_ = kind == KIND_FLOOR;
        const uint32_t tex_id = (wall ? texw : floor_ ? (texw >> 8) : (texw >> 16)) & 255u;
        const TexLds &T = tex[tex_id];
        const float *lit = fc + (wall ? FC_LIT_WALL + 3 * side : floor_ ? FC_LIT_FLOOR : FC_LIT_CEIL);
        // plane of the surface: axis (0 x, 1 y, 2 z) and coordinate
        const bool is_x = wall && !(side & 1u), is_z = wall && (side & 1u);
        const float wall_plane = side == 0u ? rect.y : side == 2u ? rect.x : side == 3u ? rect.w : rect.z;
        const int axis = is_x ? 0 : (is_z ? 2 : 1);
        const float plane = wall ? wall_plane : (floor_ ? 0.0f : height);
        // s = (a - u_org) * u_sgn * sc_s for walls, a * sc_s for floor / ceiling
        const float u_org = wall ? u_org_w : 0.0f;
        const float sc_s = T.sc_s, sc_t = T.sc_t;
        const float u_scale = (wall && ((texw >> (24 + side)) & 1u)) ? -sc_s : sc_s;
        // centre ray and the rays through the +1 pixel neighbours, recomputed here rather than kept live
        // across the sample loop (registers are what limits occupancy)
        float dc[3], dx[3], dy[3];
        make_ray(cam, cx, cy, dc);
        neighbour_rays(dc, dx, dy);
        float s0, t0, s1, t1, s2, t2;
        bool valid;
        const bool hit0 = plane_texcoord(axis, plane, u_org, u_scale, sc_t, cam.eye, dc, s0, t0);
        const bool v1 = plane_texcoord(axis, plane, u_org, u_scale, sc_t, cam.eye, dx, s1, t1);
        const bool v2 = plane_texcoord(axis, plane, u_org, u_scale, sc_t, cam.eye, dy, s2, t2);
        valid = v1 && v2;
        if (!INTERIOR && !hit0) {   // centre ray misses the plane (edge pixels only): shade at the sample's own hit point
            float dv[3];
            make_ray(cam, cx + c_sample_x[first_k], cy + c_sample_y[first_k], dv);
            plane_texcoord(axis, plane, u_org, u_scale, sc_t, cam.eye, dv, s0, t0);
            valid = false; s1 = s2 = s0; t1 = t2 = t0;
        }
        float texel[3];
        sample_texture(texels, T, s0, t0, s1, t1, s2, t2, valid, texel);
#pragma unroll
        for (int q = 0; q < 3; q++) col[q] = lit[q] * (texel[q] * (1.0f / 255.0f));
    }

    // acc = sum over the 8 samples of their colour; t_s0 / kind of sample 0 for the depth map
    // MEAN: acc already holds the mean of the 8 samples (interior pixels: 8 c / 8 == c exactly)
    template <bool MEAN>
    __device__ __forceinline__ void write_pixel(int px, int py, const float *acc, bool s0_drawn, float t_s0) const {
        uint8_t out[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float v = MEAN ? acc[q] : acc[q] * 0.125f;
            v = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);              // clamp
            out[q] = (uint8_t)(unsigned)(v * 255.0f + 0.5f);         // RGBA32F -> unorm8 resolve: floor(255 v + 1/2), v >= 0
        }
        if (layout == MWB_LAYOUT_HWC) {
            uint8_t *p = fb + (py * W + px) * 3;
            p[0] = out[0]; p[1] = out[1]; p[2] = out[2];
        } else {
            uint8_t *p = fb + (px * H + py);
            const int plane = W * H;
            p[0] = out[0]; p[plane] = out[1]; p[2 * plane] = out[2];
        }
        if (depth) {
            int z16 = 65535;   // DEPTH_COMPONENT16 of sample 0; cleared to 1.0 where nothing was drawn
            if (s0_drawn) {
                float zndc = zA - zB / t_s0;
                float dd = 0.5f * zndc + 0.5f;
                int z = (int)floorf(dd * 65535.0f + 0.5f);
                z16 = z < 0 ? 0 : (z > 65535 ? 65535 : z);
            }
            // get_depth_map, opengl.py:362-367 in float32
            float dm = (float)z16 / 65535.0f;
            float clip_z = (dm - 0.5f) * 2.0f;
            depth[py * depth_stride + px] = (float)(-2.0 * 100.0 * 0.04) / (clip_z * (float)(100.0 - 0.04) - (float)(100.0 + 0.04));
        }
    }

    // the full 8-sample path for one pixel.  skip: the first `skip` portal crossings are common to the pixel's four corner
    // rays, so every ray inside the pixel makes them too (each portal is convex: the rays through it form a convex cone
    // that holds the corners, hence their hull) - the eight traversals start in the room behind them, found by walking
    // the centre ray.  The traversal computes every distance from the eye, never from the previous room, so the
    // results are bit-identical to a start in the eye's room (guarded by test_fast_path_equals_full_sample_path's
    // MWB_DEBUG=8 leg).
    // The list of (ray, mesh) pairs of a round, walked by the whole wave: a lane takes the next pair off the list (an LDS cursor) as
    // soon as it has finished its own - walks differ in length by an order of magnitude (a ray grazing a ball's outline visits 30
    // nodes, one that misses its box two), so with 64 pairs started and finished together the wave ran as long as its longest walk
    // (40 % of the lanes busy).  A pair's result goes into its ray's slot with a 64-bit atomic minimum (t, then non-mesh before mesh,
    // then slot, then triangle: the drawing order's tie rules, whatever order the pairs finish in).
    __device__ __forceinline__ void walk_pairs(int n_tasks, int half) const {
        const f4n *gd = (const f4n *)mesh_data;
        const f4n *nodes = gd, *tris = gd;
        float lo[3] = {0, 0, 0}, ld[3] = {0, 0, 0}, inv[3] = {0, 0, 0}, th = 0.0f;
        int node = 0, n_nodes = 0, best = -1, leaf_first = 0, leaf_cnt = 0, bi = 0, slot_idx = 0;
        bool have = false, more = true;
        unsigned c_visits = 0, c_tris = 0, c_iters = 0, c_walks = 0;
        for (;;) {
            const bool tasks_left = mb_count[1] < n_tasks;   // wave-uniform (one LDS word): is there anything to refill idle lanes with
            // (a) lanes without a pair in progress: hand in the finished one, take the next
            while (leaf_cnt == 0 && node >= n_nodes && (have || more)) {   // (a lane waiting at its mesh's last leaf is not done)
                if (have) {
                    if (best >= 0) atomicMin(&mb_slots[slot_idx], ((unsigned long long)(uint32_t)__float_as_int(th) << 32) | 0x80000000u | ((uint32_t)bi << 24) | (uint32_t)best);
                    have = false;
                }
                const int ti = atomicAdd(mb_count + 1, 1);
                if (ti >= n_tasks) { more = false; break; }
                const uint32_t tk = mb_tasks[ti];
                const int src = tk & 63u, kk = (tk >> 6) & 3u;
                bi = tk >> 8;
                slot_idx = kk * WAVE + src;
                const uint32_t sp = mb_pix[src];
                const float scx = (float)(sp & 0xFFFFu) + 0.5f, scy = (float)(H - 1 - (int)(sp >> 16)) + 0.5f;
                float dv[3];
                const int k = half * MB_HALF + kk;
                make_ray(cam, scx + c_sample_x[k], scy + c_sample_y[k], dv);
                th = __int_as_float((int)(mb_slots[slot_idx] >> 32));   // nothing farther matters (another pair's result may already be in)
                const float *blk = fc + bi * FC_BOX_STRIDE;
                mesh_local_dir(blk, dv, ld);
                lo[0] = blk[FC_BOX_LO]; lo[1] = blk[FC_BOX_LO + 1]; lo[2] = blk[FC_BOX_LO + 2];
                // the node boxes are padded by 1e-4 of the mesh (host): a 1-ulp reciprocal cannot lose a real hit
                inv[0] = __builtin_amdgcn_rcpf(ld[0]); inv[1] = __builtin_amdgcn_rcpf(ld[1]); inv[2] = __builtin_amdgcn_rcpf(ld[2]);
                const uint4 md = mdesc[__float_as_int(blk[FC_LIT_BOX + FE_MESH_GEOM])];
                // the threading that visits the nearer child first for this ray's direction (bit a: component a negative)
                const uint32_t oct = md.w == 8u ? ((ld[0] < 0.0f ? 1u : 0u) | (ld[1] < 0.0f ? 2u : 0u) | (ld[2] < 0.0f ? 4u : 0u)) : 0u;
                nodes = gd + md.x + (size_t)(oct * 2u * md.z); tris = gd + md.y; n_nodes = (int)md.z;
                node = 0; best = -1; have = true;
                c_walks++;
            }
            if (!__any(node < n_nodes || leaf_cnt)) break;
            // (b) node steps only, two records per fetch (a descent costs no second round trip).  Most walks end without ever
            // reaching a leaf (a ray through a ball's gate sphere that misses the ball): the loop is left as soon as a quarter of
            // the lanes sit idle with pairs still on the list (back to (a)), or enough lanes wait at a leaf to make the leaf phase -
            // eight triangle tests for the whole wave - worth its ~700 instructions.
            for (;;) {
                const bool stepping = leaf_cnt == 0 && node < n_nodes;
                const unsigned long long ms = __ballot(stepping);
                if (!ms) break;
                const int n_leaf = __popcll(__ballot(leaf_cnt != 0));
                const int n_idle = WAVE - __popcll(ms) - n_leaf;
                if ((tasks_left && n_idle >= WAVE / 4) || n_leaf >= WAVE / 2) break;
                if (!stepping) continue;
                c_iters++;
                const int nx = node + 1 < n_nodes ? node + 1 : node;
                const f4n a = nodes[2 * node], bb = nodes[2 * node + 1], a2 = nodes[2 * nx], b2 = nodes[2 * nx + 1];
                auto visit = [&](const f4n &a, const f4n &bb) {
                    c_visits++;
                    float t0 = (a.x - lo[0]) * inv[0], t1 = (bb.x - lo[0]) * inv[0];
                    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                    t0 = (a.y - lo[1]) * inv[1]; t1 = (bb.y - lo[1]) * inv[1];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    t0 = (a.z - lo[2]) * inv[2]; t1 = (bb.z - lo[2]) * inv[2];
                    tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                    const uint32_t fcnt = (uint32_t)__float_as_int(bb.w);
                    const bool hit = tn <= tf * 1.00001f && tf > 0.0f && tn <= th;
                    const int cnt = (int)(fcnt >> 24);
                    const bool down = hit && cnt == 0;
                    node = down ? node + 1 : __float_as_int(a.w);
                    if (hit && cnt) { leaf_first = (int)(fcnt & 0xFFFFFFu); leaf_cnt = cnt; }
                    return down;
                };
                if (visit(a, bb)) visit(a2, b2);
            }
            // (c) leaves: spec arithmetic; triangle q + 1 requested before triangle q is tested.  Postponed while few lanes wait and
            // others can still move on.
            const bool can_step = __any(leaf_cnt == 0 && node < n_nodes);
            const int n_at_leaf = __popcll(__ballot(leaf_cnt != 0));
            if (leaf_cnt && (n_at_leaf >= WAVE / 2 || !can_step || !tasks_left)) {
                c_iters++;
                f4n p0 = tris[3 * leaf_first], p1 = tris[3 * leaf_first + 1], p2 = tris[3 * leaf_first + 2];
                for (int q = 0; q < leaf_cnt; q++) {
                    const f4n q0 = p0, q1 = p1, q2 = p2;
                    if (q + 1 < leaf_cnt) { p0 = tris[3 * (leaf_first + q + 1)]; p1 = tris[3 * (leaf_first + q + 1) + 1]; p2 = tris[3 * (leaf_first + q + 1) + 2]; }
                    const float4 r0 = make_float4(q0.x, q0.y, q0.z, q0.w), r1 = make_float4(q1.x, q1.y, q1.z, q1.w), r2 = make_float4(q2.x, q2.y, q2.z, q2.w);
                    float t, u, v, det;
                    c_tris++;
                    if (mesh_tri(lo, ld, r0, r1, r2, true, t, u, v, det)) {
                        const int idx = __float_as_int(r2.y);
                        // t == th with no hit of this walk yet: kept - th may be another pair's result for the same ray, and the
                        // merge decides
                        if (t < th || (t == th && (best < 0 || idx < best))) { th = t; best = idx; }
                    }
                }
                leaf_cnt = 0;
            }
        }
        if ((exp_flags & 4) && dbg_counters) {
            atomicAdd(dbg_counters + 0, (unsigned long long)c_walks); atomicAdd(dbg_counters + 1, (unsigned long long)c_walks);
            atomicAdd(dbg_counters + 2, (unsigned long long)c_visits); atomicAdd(dbg_counters + 3, (unsigned long long)c_tris);
            unsigned mx = c_iters;
            for (int o = 32; o; o >>= 1) { const unsigned v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
            if ((threadIdx.x & 63) == 0) { atomicAdd(dbg_counters + 4, (unsigned long long)mx); atomicAdd(dbg_counters + 5, 1ull); }
        }
    }

    // Entity tasks: a batch of up to 64 pixels out of the mesh-pixel queue (lane = pixel; valid: this lane holds one).  pixel_full's
    // result, bit for bit, with the mesh walks REDISTRIBUTED over the wave: a pixel inside a ball's outline has eight walks to do and
    // its neighbour outside none, so with one pixel per lane a wave's loop ran as long as its busiest lane (a third of the lanes busy).
    // Per half of the samples (4): every lane traces rooms, boxes and frames for its own pixel's samples and leaves the nearest
    // surface in LDS as an ordered word (t, then non-mesh before mesh, then slot, then triangle - the drawing order's tie rules);
    // every (sample ray, mesh) pair whose gate sphere the ray passes goes to a list; the list is walked 64 pairs at a time by
    // whichever lanes (walk_pair), results merged with a 64-bit atomic minimum; the lanes then read their own pixels' results back.
    __device__ __forceinline__ void pixels_mesh(int px, int py, int skip, bool valid) const {
        const int lane = threadIdx.x & (WAVE - 1);
        const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
        int start_room = cam_room;
        if (valid && skip > 0) {
            float dc[3];
            make_ray(cam, cx, cy, dc);
            const int r = walk_rooms(rooms, cam_room, cam.eye, dc, skip);
            start_room = r >= 0 ? r : cam_room;
        }
        const uint32_t my_boxes = valid && boxes_in_view ? item_boxes(px, py) : 0u;
        uint32_t any_boxes = 0;
#pragma unroll
        for (int bi = 0; bi < NBOX; bi++) any_boxes |= __ballot((my_boxes >> bi) & 1u) ? 1u << bi : 0u;
        mb_pix[lane] = (uint32_t)px | ((uint32_t)py << 16);
        const int task_cap = (exp_flags & 32) ? 8 : MB_TASKS;   // MWB_EXP bit 5: a list of 8 pairs, so that tests reach the overflow path
        uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0, meta = 0, key_s0 = 0;
        float t_s0 = INFINITY;
        float acc[3] = {0, 0, 0};
#pragma unroll 1
        for (int half = 0; half < 8 / MB_HALF; half++) {
            if (lane == 0) { mb_count[0] = 0; mb_count[1] = 0; }   // pairs listed, pairs taken
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t knm[MB_HALF];
#pragma unroll
            for (int kk = 0; kk < MB_HALF; kk++) {
                const int k = half * MB_HALF + kk;
                float dv[3], th = INFINITY;
                uint32_t path, key = 0;
                make_ray(cam, cx + c_sample_x[k], cy + c_sample_y[k], dv);
                if (valid) key = trace_from<false>(start_room, dv, th, path);
                const float dd = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
#pragma unroll 1
                for (uint32_t bm = any_boxes & ~mesh_slots; bm; bm &= bm - 1u) {   // boxes and frames, as in pixel_full
                    const int bi = __builtin_ctz(bm);
                    const float *fb_ = fc + bi * FC_BOX_STRIDE;
                    const float cc = fb_[FC_CULL_CC];
                    const float b = dv[0] * fb_[FC_CULL_OC] + dv[1] * fb_[FC_CULL_OC + 1] + dv[2] * fb_[FC_CULL_OC + 2];
                    if (((my_boxes >> bi) & 1u) && (cc <= 0.0f || (b > 0.0f && b * b >= dd * cc))) {
                        float tb;
                        const float hxk = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fb_[FC_BOX_HX])));
                        if (hxk == -2.0f) {
                            const int code = trace_frame(fb_, dv, tb);
                            if (code >= 0 && tb < th) { key = KIND_FRAME | ((uint32_t)bi << 3) | ((uint32_t)code << 8); th = tb; }
                        } else {
                            int face = trace_box(fb_, dv, tb);
                            if (face >= 0 && tb < th) { key = MAKE_KEY(KIND_BOX, face, bi); th = tb; }
                        }
                    }
                }
                knm[kk] = key;
                mb_slots[kk * WAVE + lane] = (unsigned long long)(uint32_t)__float_as_int(th) << 32;
                if (!(exp_flags & 1)) {
                    for (uint32_t c = my_boxes & mesh_slots; c; c &= c - 1u) {   // the gate sphere of every candidate mesh
                        const int bi = __builtin_ctz(c);
                        const float *blk = fc + bi * FC_BOX_STRIDE;
                        const float cc = blk[FC_CULL_CC];
                        const float b = dv[0] * blk[FC_CULL_OC] + dv[1] * blk[FC_CULL_OC + 1] + dv[2] * blk[FC_CULL_OC + 2];
                        if (!(cc <= 0.0f || (b > 0.0f && b * b >= dd * cc))) continue;
                        const int slot = atomicAdd(mb_count, 1);
                        if (slot < task_cap) mb_tasks[slot] = (uint16_t)(lane | (kk << 6) | (bi << 8));
                        else {   // list full (dozens of meshes behind one pixel): walked on the spot, by this lane
                            float t; int tri;
                            // (every lane must take part in walk_pair's wave-wide loop: not here - a plain per-lane walk)
                            uint32_t kk_key = 0; float th_l = __int_as_float((int)(mb_slots[kk * WAVE + lane] >> 32));
                            walk_meshes(1u << bi, dv, dd, th_l, kk_key);
                            if (kk_key) atomicMin(&mb_slots[kk * WAVE + lane], ((unsigned long long)(uint32_t)__float_as_int(th_l) << 32) | 0x80000000u | ((uint32_t)bi << 24) | (kk_key >> 8));
                            (void)t; (void)tri;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int n_tasks = min(*mb_count, task_cap);
            if (n_tasks > 0 && !(exp_flags & 16)) walk_pairs(n_tasks, half);   // MWB_EXP bit 4: pairs listed but not walked (timing experiments)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < MB_HALF; kk++) {   // my pixel's samples: the distinct surfaces in order of first appearance (pixel_full)
                const int k = half * MB_HALF + kk;
                const unsigned long long sl = mb_slots[kk * WAVE + lane];
                const uint32_t lw = (uint32_t)sl;
                const float th = __int_as_float((int)(sl >> 32));
                const uint32_t key = (lw & 0x80000000u) ? (KIND_MESH | (((lw >> 24) & 31u) << 3) | ((lw & 0xFFFFFFu) << 8)) : knm[kk];
                if (k == 0) { key_s0 = key; t_s0 = th; }
                const bool h0 = (meta & 0xFu) && key == k0, h1 = (meta & 0xF0u) && key == k1;
                const bool h2 = (meta & 0xF00u) && key == k2, h3 = (meta & 0xF000u) && key == k3;
                meta += (h0 ? 1u : 0u) + (h1 ? 0x10u : 0u) + (h2 ? 0x100u : 0u) + (h3 ? 0x1000u : 0u);
                if (!(h0 || h1 || h2 || h3)) {
                    if (!(meta & 0xFu)) { k0 = key; meta |= 1u | ((uint32_t)k << 16); }
                    else if (!(meta & 0xF0u)) { k1 = key; meta |= 0x10u | ((uint32_t)k << 19); }
                    else if (!(meta & 0xF00u)) { k2 = key; meta |= 0x100u | ((uint32_t)k << 22); }
                    else if (!(meta & 0xF000u)) { k3 = key; meta |= 0x1000u | ((uint32_t)k << 25); }
                    else if (valid) {
                        float col[3];
                        shade<false>(key, k, cx, cy, col);
                        acc[0] += col[0]; acc[1] += col[1]; acc[2] += col[2];
                    }
                }
            }
        }
        if (!valid) return;
#pragma unroll 1
        for (int it = 0; it < 4 && (meta & 0xFu); it++) {
            float col[3];
            shade<false>(k0, (meta >> 16) & 7u, cx, cy, col);
            const float cnt = (float)(meta & 0xFu);
#pragma unroll
            for (int q = 0; q < 3; q++) acc[q] += cnt * col[q];
            k0 = k1; k1 = k2; k2 = k3;
            meta = ((meta & 0xFFFFu) >> 4) | ((meta >> 19) << 16);
        }
        write_pixel<false>(px, py, acc, (key_s0 & 7u) != KIND_SKY, t_s0);
    }

    __device__ __forceinline__ void pixel_full(int px, int py, int skip) const {
        const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
        int start_room = cam_room;
        if (skip > 0) {
            float dc[3];
            make_ray(cam, cx, cy, dc);
            const int r = POLY ? walk_rooms_poly(rooms, cam_room, cam.eye, dc, skip) : walk_rooms(rooms, cam_room, cam.eye, dc, skip);
            start_room = r >= 0 ? r : cam_room;
        }
        // boxes that can touch this pixel's item at all, and their union over the batch (scalar): most batches see none
        const uint32_t my_boxes = boxes_in_view ? item_boxes(px, py) : 0u;
        uint32_t any_boxes = boxes_in_view ? 1u : 0u;
        if (NBOX > 1) {
            any_boxes = 0;
#pragma unroll
            for (int bi = 0; bi < NBOX; bi++) any_boxes |= __ballot((my_boxes >> bi) & 1u) ? 1u << bi : 0u;
        }
        // distinct surfaces among the 8 coverage samples, in order of first appearance:
        // up to 4 slots (key, count, first sample); a 5th distinct surface is shaded on the spot
        uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        uint32_t meta = 0;   // four 4-bit sample counts (bits 0-15) and four 3-bit first-sample indices (bits 16-27)
        uint32_t key_s0 = 0;
        float t_s0 = INFINITY;
        float acc[3] = {0, 0, 0};
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            float dv[3], th;
            uint32_t path;
            make_ray(cam, cx + c_sample_x[k], cy + c_sample_y[k], dv);
            uint32_t key = trace_from<false>(start_room, dv, th, path);
            // conservative bounding-sphere cull, then the exact slab test; boxes in entity order, a later
            // box wins only when strictly nearer
            const float dd = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
            if (!any_boxes) {
            } else if (NBOX == 1) {   // cull constants pinned to scalar registers
#pragma unroll
                for (int bi = 0; bi < NBOX; bi++) {
                    const float b = dv[0] * cull_oc[bi][0] + dv[1] * cull_oc[bi][1] + dv[2] * cull_oc[bi][2];
                    if (cull_cc[bi] <= 0.0f || (b > 0.0f && b * b >= dd * cull_cc[bi])) {
                        float tb;
                        int face = trace_box(fc + bi * FC_BOX_STRIDE, dv, tb);
                        if (face >= 0 && tb < th) { key = MAKE_KEY(KIND_BOX, face, bi); th = tb; }
                    }
                }
            } else {
#pragma unroll 1
                for (uint32_t bm = NBOX > MWB_MAX_BOXES ? any_boxes & ~mesh_slots : any_boxes; bm; bm &= bm - 1u) {   // a real loop: one copy of the slab test, constants from LDS
                    const int bi = __builtin_ctz(bm);                 // (entity order kept: a later box wins only when strictly nearer)
                    const float *fb_ = fc + bi * FC_BOX_STRIDE;
                    const float cc = fb_[FC_CULL_CC];
                    const float b = dv[0] * fb_[FC_CULL_OC] + dv[1] * fb_[FC_CULL_OC + 1] + dv[2] * fb_[FC_CULL_OC + 2];
                    if (((my_boxes >> bi) & 1u) && (cc <= 0.0f || (b > 0.0f && b * b >= dd * cc))) {
                        float tb;
                        int ekind = 0;   // 0 box, 1 mesh, 2 frame (entity tasks; workgroup-uniform per slot)
                        if constexpr (NBOX > MWB_MAX_BOXES) {
                            const float hxk = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fb_[FC_BOX_HX])));
                            ekind = hxk == -1.0f ? 1 : hxk == -2.0f ? 2 : 0;
                        }
                        if (ekind == 1) {   // meshes: walked below, every lane through ITS OWN candidates at the same time
                        } else if (ekind == 2) {
                            const int code = trace_frame(fb_, dv, tb);
                            if (code >= 0 && tb < th) { key = KIND_FRAME | ((uint32_t)bi << 3) | ((uint32_t)code << 8); th = tb; }
                        } else {
                            int face = trace_box(fb_, dv, tb);
                            if (face >= 0 && tb < th) { key = MAKE_KEY(KIND_BOX, face, bi); th = tb; }
                        }
                    }
                }
                // (entity tasks: no mesh can touch a pixel that comes this way - the pixels a mesh may cover, by the pixel-inflated gate
                // of emit(), are batched apart and go through pixels_mesh)
            }
            if (k == 0) { key_s0 = key; t_s0 = th; }
            const bool h0 = (meta & 0xFu) && key == k0, h1 = (meta & 0xF0u) && key == k1;
            const bool h2 = (meta & 0xF00u) && key == k2, h3 = (meta & 0xF000u) && key == k3;
            meta += (h0 ? 1u : 0u) + (h1 ? 0x10u : 0u) + (h2 ? 0x100u : 0u) + (h3 ? 0x1000u : 0u);
            if (!(h0 || h1 || h2 || h3)) {
                if (!(meta & 0xFu)) { k0 = key; meta |= 1u | ((uint32_t)k << 16); }
                else if (!(meta & 0xF0u)) { k1 = key; meta |= 0x10u | ((uint32_t)k << 19); }
                else if (!(meta & 0xF00u)) { k2 = key; meta |= 0x100u | ((uint32_t)k << 22); }
                else if (!(meta & 0xF000u)) { k3 = key; meta |= 0x1000u | ((uint32_t)k << 25); }
                else {
                    float col[3];
                    shade<false>(key, k, cx, cy, col);
                    acc[0] += col[0]; acc[1] += col[1]; acc[2] += col[2];
                }
            }
        }
#pragma unroll 1
        for (int it = 0; it < 4 && (meta & 0xFu); it++) {
            float col[3];
            shade<false>(k0, (meta >> 16) & 7u, cx, cy, col);
            const float cnt = (float)(meta & 0xFu);
#pragma unroll
            for (int q = 0; q < 3; q++) acc[q] += cnt * col[q];
            k0 = k1; k1 = k2; k2 = k3;
            meta = ((meta & 0xFFFFu) >> 4) | ((meta >> 19) << 16);
        }
        write_pixel<false>(px, py, acc, (key_s0 & 7u) != KIND_SKY, t_s0);
    }

    // a pixel whose 8 samples provably see one room surface: one shade, colour x 8 / 8
    __device__ __forceinline__ void pixel_interior(int px, int py, uint32_t key) const {
        const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
        float col[3];
        shade<true>(key, 0, cx, cy, col);
        float t_s0 = 1.0f;
        if (depth) {   // sample 0's ray meets the known surface at the distance the traversal would report
            float dv[3];
            make_ray(cam, cx + c_sample_x[0], cy + c_sample_y[0], dv);
            const uint32_t kind = key & 7u, side = (key >> 3) & 7u;
            if constexpr (POLY) {
                const float *r = rooms + __umul24(key >> 6, MWB_POLY_ROOM_WORDS);
                const float *ed = r + PW_EDGE0 + __umul24(side & 3u, PW_EDGE_WORDS);
                const float den = fmaf(ed[5], dv[2], ed[4] * dv[0]);
                const float num = fmaf(ed[5], ed[1] - cam.eye[2], ed[4] * (ed[0] - cam.eye[0]));
                const float plane = kind == KIND_CEIL ? fabsf(r[PW_HEIGHT]) : 0.0f;
                t_s0 = kind == KIND_WALL ? num / den : (plane - cam.eye[1]) * (1.0f / dv[1]);
                write_pixel<true>(px, py, col, true, t_s0);
                return;
            }
            const float *r = rooms + __umul24(key >> 6, MWB_ROOM_WORDS);
            const float4 rect = *(const float4 *)(r + RW_MINX);
            const bool wall = kind == KIND_WALL;
            const bool is_x = wall && (side == 0u || side == 2u), is_z = wall && (side == 1u || side == 3u);
            const float wall_plane = side == 0u ? rect.y : side == 2u ? rect.x : side == 3u ? rect.w : rect.z;
            const float plane = wall ? wall_plane : (kind == KIND_CEIL ? r[RW_HEIGHT] : 0.0f);
            const float od = is_x ? dv[0] : (is_z ? dv[2] : dv[1]);
            const float oo = is_x ? cam.eye[0] : (is_z ? cam.eye[2] : cam.eye[1]);
            t_s0 = (plane - oo) * (1.0f / od);
        }
        write_pixel<true>(px, py, col, true, t_s0);
    }
};

#define TILE_CX 16   // corner grid of one wave pass: 16 x 4 corners; marching down a strip it classifies 15 x 4 pixels
#define TILE_CY 4
#define QUEUE_CAP 128
#define ITEM_RES_BYTES(W) ((((W) + TILE_CX - 2) / (TILE_CX - 1)) * 4 * 16)   // n_strips x 4 quarters x uint4

// LDS -> HBM copy of the byte range [begin, end) of the frame with the widest vectors its alignment allows
template <int THREADS>
__device__ __forceinline__ void copy_frame_range(uint8_t *__restrict__ dst, const uint8_t *__restrict__ fb, int begin, int end, int tid) {
    if (((begin | end) & 15) == 0) {
        const uint4 *s4 = (const uint4 *)fb;
        uint4 *d4 = (uint4 *)dst;
        for (int i = begin / 16 + tid; i < end / 16; i += THREADS) d4[i] = s4[i];
    } else if (((begin | end) & 3) == 0) {
        const uint32_t *s1 = (const uint32_t *)fb;
        uint32_t *d1 = (uint32_t *)dst;
        for (int i = begin / 4 + tid; i < end / 4; i += THREADS) d1[i] = s1[i];
    } else {
        for (int i = begin + tid; i < end; i += THREADS) dst[i] = fb[i];
    }
}

// Renders one env with the whole workgroup (called once per workgroup, or per list entry on the side stream).
// part < 0: the whole frame; part 0 / 1: one half of it (the last envs of a bulk launch are cut in two so that the
// launch drains in units of half a workgroup time) - the upper / lower rows for HWC frames, the left / right
// strips for CWH ones, so that a half's bytes are one (three) contiguous run(s).
// TILED: the workgroup renders the tile [tx0, tx0 + tw) x [ty0, ty0 + th) of a larger view of d.W x d.H pixels (mwb_render_view: the
// reference's 800 x 600 human view, or any observation size whose frame does not fit LDS): everything below works in tile-local
// pixel coordinates; only the camera's window mapping carries the offset - (2 (wx + ox) - W) / W = (2 wx - (W - 2 ox)) / W, exact
// in float32 - and the tile's rows go to their places in the big frame.
template <int THREADS, int NBOX, bool LOOPED, bool POLY, bool TILED = false>
__device__ __forceinline__ void render_env(const MwbDev &d, const int e, const int part, unsigned char *smem, const int tx0 = 0,
                                           const int ty0 = 0, const int tw = 0, const int th = 0) {
    // LOOPED (the body sits in a loop over the regenerated-env list): make the lane id opaque to the optimiser so that
    // nothing derived from it is hoisted out of that loop and kept alive across whole frames (200 -> 76 B/lane of scratch)
    int tid = threadIdx.x;
    if (LOOPED) asm volatile("" : "+v"(tid));
    const int W = TILED ? tw : d.W, H = TILED ? th : d.H;
    int n_rooms = d.n_rrooms[e];
    if (n_rooms < 0) n_rooms = 0;
    float *rooms = (float *)smem;
    size_t off = ((size_t)d.R_max * d.room_words * 4 + 15) & ~(size_t)15;
    float *fc = (float *)(smem + off); off += (size_t)d.frame_words * 4;
    TexLds *tex = (TexLds *)(smem + off); off += sizeof(TexLds) * d.n_tex;
    int *cam_room_s = (int *)(smem + off); off += 16 + 2 * (THREADS / WAVE) * sizeof(int);   // + leftover counts
    uint16_t *queues = (uint16_t *)(smem + off); off += (THREADS / WAVE) * QUEUE_CAP * sizeof(uint16_t);
    uint32_t *ikeys = (uint32_t *)(smem + off); off += (THREADS / WAVE) * QUEUE_CAP * sizeof(uint32_t);
    uint16_t *ipix = (uint16_t *)(smem + off); off += (THREADS / WAVE) * QUEUE_CAP * sizeof(uint16_t);
    uint4 *item_res = (uint4 *)(smem + off); off += (size_t)ITEM_RES_BYTES(W);   // per work item: uniform rows + their key
    uint8_t *fb = smem + off;   // the frame is assembled in LDS and leaves as 16-byte coalesced stores
    // entity tasks: a third queue per wave for the 8-sample pixels a mesh may cover, and its leftover counts (behind the frame)
    uint16_t *mqueues = (uint16_t *)(smem + ((off + (size_t)W * H * 3 + 15) & ~(size_t)15));
    int *mleft = (int *)(mqueues + (THREADS / WAVE) * MQ_CAP);
    uint4 *mdesc = (uint4 *)(mleft + 4);   // entity tasks: [MWB_NUM_MESHES], see RenderCtx::mdesc
    uint8_t *mb_base = (uint8_t *)(mdesc + MWB_NUM_MESHES) + (size_t)(tid / WAVE) * MB_WAVE_BYTES;   // this wave's batch scratch (pixels_mesh)

    {   // stage the room table, the frame constants and the texture descriptors
        const int rw = POLY ? MWB_POLY_ROOM_WORDS : MWB_ROOM_WORDS;
        const float4 *src = (const float4 *)(d.rooms + (size_t)e * d.R_max * rw);
        float4 *dst = (float4 *)rooms;
        for (int i = tid; i < n_rooms * (rw / 4); i += THREADS) dst[i] = src[i];
        const float *fsrc = d.frame + (size_t)e * d.frame_words;
        for (int i = tid; i < d.frame_words; i += THREADS) fc[i] = fsrc[i];
        const uint32_t *tsrc = (const uint32_t *)d.tex_desc;
        uint32_t *tdst = (uint32_t *)tex;
        for (int i = tid; i < (int)(sizeof(TexLds) / 4) * d.n_tex; i += THREADS) tdst[i] = tsrc[i];
        if (tid == 0) { cam_room_s[0] = 0x7fffffff; cam_room_s[1] = 0; }   // eye room (atomicMin), work-item counter
        if constexpr (NBOX > MWB_MAX_BOXES)
            if (tid < MWB_NUM_MESHES) { const MwbMeshDesc &m = d.mesh_desc[tid]; mdesc[tid] = make_uint4(m.node_off, m.tri_off, (uint32_t)m.n_nodes, (uint32_t)m.n_orders); }
    }
    __syncthreads();
    RenderCtx<NBOX, POLY> ctx;
    ctx.rooms = rooms; ctx.fc = fc; ctx.tex = tex; ctx.texels = d.texels; ctx.fb = fb;
    ctx.mesh_desc = d.mesh_desc; ctx.mesh_data = d.mesh_data; ctx.mdesc = mdesc;
    ctx.mb_slots = (unsigned long long *)mb_base; ctx.mb_tasks = (uint16_t *)(mb_base + MB_HALF * WAVE * 8);
    ctx.mb_pix = (uint32_t *)(mb_base + MB_HALF * WAVE * 8 + MB_TASKS * 2); ctx.mb_count = (int *)(mb_base + MB_HALF * WAVE * 8 + MB_TASKS * 2 + WAVE * 4);
    ctx.exp_flags = d.exp_flags; ctx.mesh_slots = 0; ctx.dbg_counters = d.dbg_counters;
    if constexpr (NBOX > MWB_MAX_BOXES) {   // which slots hold a mesh in this frame
        uint32_t ms = 0;
        for (int bi = 0; bi < d.n_boxes; bi++) ms |= fc[bi * FC_BOX_STRIDE + FC_BOX_HX] == -1.0f ? 1u << bi : 0u;
        ctx.mesh_slots = (uint32_t)__builtin_amdgcn_readfirstlane((int)ms);
    }
    ctx.depth = d.want_depth ? d.depth + (size_t)e * d.W * d.H + (TILED ? (size_t)ty0 * d.W + tx0 : 0) : nullptr;
    ctx.depth_stride = TILED ? d.W : W;
    ctx.n_rooms = n_rooms; ctx.W = W; ctx.H = H; ctx.layout = d.layout;
    Cam &cam = ctx.cam;
    // frame constants are workgroup-uniform: pin them to scalar registers
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
    for (int k = 0; k < 3; k++) { cam.eye[k] = uni(fc[FC_EYE + k]); cam.F[k] = uni(fc[FC_F + k]); cam.S[k] = uni(fc[FC_S + k]); cam.U[k] = uni(fc[FC_U + k]); }
    cam.TW = uni(fc[FC_TW]); cam.TH = uni(fc[FC_TH]);
    cam.Wf = (float)W; cam.Hf = (float)H; cam.invW = 1.0f / (float)W; cam.invH = 1.0f / (float)H;
    if (TILED) {   // window x of tile column c = tx0 + c; window y (up) of tile row r (from the top) = (d.H - ty0 - th) + (th - r)
        cam.Wf = (float)(d.W - 2 * tx0); cam.Hf = (float)(d.H - 2 * (d.H - ty0 - th));
        cam.invW = 1.0f / (float)d.W; cam.invH = 1.0f / (float)d.H;
    }
    // room containing the eye: first (lowest index) rectangle that holds it, bounds inclusive.  Every wave finds it for
    // itself (64 rooms per sweep, lowest set bit of the first non-empty ballot): no LDS atomic, no barrier
    ctx.cam_room = -1;
    for (int base = 0; base < n_rooms; base += WAVE) {
        const int i = base + (tid & (WAVE - 1));
        bool in = i < n_rooms;
        if (in) {
            if constexpr (POLY) {   // the eye on the inner side of every edge; never a culled room
                const float *r = rooms + i * MWB_POLY_ROOM_WORDS;
                in = !((__float_as_int(r[PW_FLAGS]) >> 8) & 1);
                for (int k = 0; k < 4; k++) {
                    const float *ed = r + PW_EDGE0 + PW_EDGE_WORDS * k;
                    if (!(fmaf(ed[5], cam.eye[2] - ed[1], ed[4] * (cam.eye[0] - ed[0])) >= 0.0f)) in = false;
                }
            } else {
                const float *r = rooms + i * MWB_ROOM_WORDS;
                in = cam.eye[0] >= r[RW_MINX] && cam.eye[0] <= r[RW_MAXX] && cam.eye[2] >= r[RW_MINZ] && cam.eye[2] <= r[RW_MAXZ];
            }
        }
        const unsigned long long m = __ballot(in);
        if (m) { ctx.cam_room = base + (__ffsll((long long)m) - 1); break; }
    }
    const float zn = 0.04f, zf = 100.0f;   // gluPerspective near / far, miniworld.py:1186-1187
    ctx.zA = (zf + zn) / (zf - zn); ctx.zB = (2.0f * zf * zn) / (zf - zn);
    float cull_cc_px[1] = {0.0f};   // sphere inflated by a pixel footprint (prep_kernel)
    if constexpr (NBOX == 1) {   // one box: its cull constants live in scalar registers; more boxes: read from LDS where used
        ctx.cull_cc[0] = uni(fc[FC_CULL_CC]);
        ctx.cull_oc[0][0] = uni(fc[FC_CULL_OC]); ctx.cull_oc[0][1] = uni(fc[FC_CULL_OC + 1]); ctx.cull_oc[0][2] = uni(fc[FC_CULL_OC + 2]);
        cull_cc_px[0] = uni(fc[FC_CULL_CC_PIXEL]);
    }
    (void)cull_cc_px;
    ctx.boxes_in_view = __builtin_amdgcn_readfirstlane(__float_as_int(fc[FC_BOX_IN_VIEW])) != 0;
    ctx.item_res = (d.debug_flags & (1 | 32 | 128)) ? nullptr : item_res;   // filled by the lattice pass below
    ctx.part_h_inv = 65536 / ((H + 3) / 4) + 1;

    // Pass structure per wave: a 16 x 4 grid of rays through PIXEL CORNERS per pass, marching down a
    // 15-pixel-wide strip; the last corner row of a pass is carried in registers, so a pass classifies
    // 15 x 4 pixels (1.07 corner rays per pixel).  If the four corner rays of a pixel reach the same
    // convex piece of a room surface through the same portal sequence, every ray inside the pixel does
    // (rooms, portals and the pieces are convex, samples lie >= 1/16 pixel inside the corners) and,
    // unless the box may intrude, the pixel is shaded once ("interior").  All other pixels are resolved
    // by the full 8-sample path.  Both kinds go through per-wave LDS queues and are processed 64 at a
    // time, so that the shading and the 8-sample path always run with dense lanes.
    // the wave index is wave-uniform: in a scalar register, so are the queue addresses derived from it (VGPRs are the budget)
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE), lane = tid % WAVE, n_waves = THREADS / WAVE;
    // queued pixels are packed as (py << wshift) | px (host checks that it fits 16 bits): no integer division
    const int wshift = 32 - __builtin_clz((unsigned)(W > 1 ? W - 1 : 1));
    const int wmask = (1 << wshift) - 1;
    // an 8-sample pixel's entry also carries, above the coordinates, how many leading portal crossings its corner rays share
    // (as many bits as 16 - the coordinate bits leave, at most 3; MWB_DEBUG bit 3: none)
    const int hshift = 32 - __builtin_clz((unsigned)(H > 1 ? H - 1 : 1));
    const int qshift = wshift + hshift, hmask = (1 << hshift) - 1;
    const int skip_max = (d.debug_flags & 8) || qshift >= 16 ? 0 : (1 << (16 - qshift > 3 ? 3 : 16 - qshift)) - 1;
    uint16_t *queue = queues + wave * QUEUE_CAP;
    uint32_t *iq_key = ikeys + wave * QUEUE_CAP;
    uint16_t *iq_pix = ipix + wave * QUEUE_CAP;
    int q_count = 0, iq_count = 0;   // wave-uniform
    uint16_t *mqueue = mqueues + wave * MQ_CAP;
    int mq_count = 0;
    const int ci = lane & (TILE_CX - 1), cj = lane / TILE_CX;
    const int n_strips = (W + TILE_CX - 2) / (TILE_CX - 1);
    // Work items = (strip, quarter of the rows): 24 at 80 x 60, 15 x 15 pixels each (16 corner rows = 4 passes,
    // none wasted).  Waves take them from an LDS counter as they become free, so a wave that drew cheap
    // (uniform) items does not idle while another one works through the edge-rich ones.
    const int part_h = (H + 3) / 4;                       // pixel rows per item
    const bool split_x = d.layout == MWB_LAYOUT_CWH;
    const int half_strips = (n_strips + 1) / 2;
    const int n_items = part < 0 ? n_strips * 4
                                 : (split_x ? (part == 0 ? half_strips : n_strips - half_strips) * 4 : n_strips * 2);
    (void)n_waves;

    // may a box intrude into an otherwise uniform pixel?  centre ray against the footprint-inflated bounding spheres
    auto box_may_touch = [&](int px, int py, uint32_t boxes /* wave-uniform: the item's */) {
        if (NBOX > 1 && !boxes) return false;
        float dc[3];
        make_ray(cam, (float)px + 0.5f, (float)(H - 1 - py) + 0.5f, dc);
        const float dd = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
        bool touch = false;
        if constexpr (NBOX == 1) {   // cull constants pinned to scalar registers
            const float b = dc[0] * ctx.cull_oc[0][0] + dc[1] * ctx.cull_oc[0][1] + dc[2] * ctx.cull_oc[0][2];
            touch = cull_cc_px[0] <= 0.0f || (b > 0.0f && b * b >= dd * cull_cc_px[0]);
        } else {
            // a real loop over the item's boxes, constants from LDS (workgroup-uniform addresses): six boxes' constants pinned
            // to scalar registers overflow the SGPR file into VGPRs and those into scratch
#pragma unroll 1
            for (uint32_t bm = boxes & ((1u << NBOX) - 1u); bm; bm &= bm - 1u) {
                const float *fb_ = fc + __builtin_ctz(bm) * FC_BOX_STRIDE;
                const float ccp = fb_[FC_CULL_CC_PIXEL];
                const float b = dc[0] * fb_[FC_CULL_OC] + dc[1] * fb_[FC_CULL_OC + 1] + dc[2] * fb_[FC_CULL_OC + 2];
                bool t_ = ccp <= 0.0f || (b > 0.0f && b * b >= dd * ccp);
                if constexpr (NBOX > MWB_MAX_BOXES) {   // a mesh: its pixel-inflated sphere AND its own pixel-inflated bounding box
                    if (__builtin_amdgcn_readfirstlane(__float_as_int(fb_[FC_BOX_HX])) == __float_as_int(-1.0f) && t_) {
                        const float *me = fb_ + FC_LIT_BOX;
                        float ld[3];
                        mesh_local_dir(fb_, dc, ld);
                        const float pad = me[FE_MESH_BPAD];
                        const float blo[3] = {-me[FE_MESH_BHX] - pad, -pad, -me[FE_MESH_BHZ] - pad}, bhi[3] = {me[FE_MESH_BHX] + pad, fb_[FC_BOX_SY] + pad, me[FE_MESH_BHZ] + pad};
                        float tn = -INFINITY, tf = INFINITY;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const float inv = 1.0f / ld[a], o_ = fb_[FC_BOX_LO + a];
                            const float t0 = (blo[a] - o_) * inv, t1 = (bhi[a] - o_) * inv;
                            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
                        }
                        t_ = tn <= tf && tf > 0.0f;
                    }
                }
                touch = touch || t_;
            }
        }
        return touch;
    };

    // a classified pixel goes to its per-wave queue; a queue that reaches 64 entries is processed at once (dense lanes)
    auto emit = [&](int px, int py, bool is_pixel, bool interior, uint32_t key, int skip, uint32_t iboxes) {
        bool edge = is_pixel && !interior;
        if constexpr (NBOX > MWB_MAX_BOXES) {
            // 8-sample pixels that a mesh may cover queue up apart: their batches walk mesh hierarchies with most lanes busy, while
            // the other batches never wait for a few lanes' walks (mixed, ~1 lane in 8 was walking: scripts/ab_mesh_phases.py)
            const bool mesh_px = edge && ctx.boxes_in_view && (iboxes & ctx.mesh_slots) && box_may_touch(px, py, iboxes & ctx.mesh_slots);
            const unsigned long long mm = __ballot(mesh_px);
            if (mesh_px) mqueue[mq_count + __popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)((skip << qshift) | (py << wshift) | px);
            mq_count += __popcll(mm);
            edge = edge && !mesh_px;
            // Mesh pixels cluster (a ball covers a few items of one or two waves' strips): batches are NOT processed as they fill up but
            // kept for the end of the frame, where the four waves share them evenly; only a queue about to overflow is drained here.
            if (mq_count > MQ_CAP - WAVE) {
                mq_count -= WAVE;
                const int q = mqueue[mq_count + lane];
                if (!(d.debug_flags & 2)) ctx.pixels_mesh(q & wmask, (q >> wshift) & hmask, q >> qshift, true);
            }
        }
        const unsigned long long em = __ballot(edge);
        if (edge) queue[q_count + __popcll(em & ((1ull << lane) - 1ull))] = (uint16_t)((skip << qshift) | (py << wshift) | px);
        q_count += __popcll(em);
        const unsigned long long im = __ballot(interior);
        if (interior) {
            const int slot = iq_count + __popcll(im & ((1ull << lane) - 1ull));
            iq_key[slot] = key & 0x0FFFFFFFu; iq_pix[slot] = (uint16_t)((py << wshift) | px);
        }
        iq_count += __popcll(im);
        if (iq_count >= WAVE) {
            iq_count -= WAVE;
            const int q = iq_pix[iq_count + lane];
            if (!(d.debug_flags & 4)) ctx.pixel_interior(q & wmask, q >> wshift, iq_key[iq_count + lane]);
        }
        if (q_count >= WAVE) {
            q_count -= WAVE;
            const int q = queue[q_count + lane];
            if (!(d.debug_flags & 2)) ctx.pixel_full(q & wmask, (q >> wshift) & hmask, q >> qshift);
        }
    };
    // Item pre-tests, all at once.  If the corner rays of a screen rectangle reach the same convex piece of a room surface
    // through the same portals, so does every ray in between (the argument made for a pixel holds for any rectangle): all
    // its pixels are interior pixels of that surface.  Tested for every 15 x 15 item and for its upper / lower 8 rows: the
    // corner rays of all items form a lattice of 16 rows (4 per quarter: top, rows - 8, 8, bottom) x (strips + 1) columns,
    // traced ONCE per frame by waves 0 and 1 (8 lattice columns = 7 strips per pass) - neighbouring items share corners,
    // and a wave that pulls an item finds the verdict in LDS instead of spending a whole wave-trace on 8 rays.
    // A uniform item skips its 4 corner passes, a uniform half leaves 2.
    if (!(d.debug_flags & (1 | 32 | 128))) {
        for (int chunk0 = 0; chunk0 < n_strips; chunk0 += 7) {
            if (wave < 2) {
                const int r = lane >> 3, k = lane & 7, q = wave * 2 + (r >> 2), rr = r & 3;
                const int row0 = q * part_h;
                const int rows = (H - row0) < part_h ? ((H - row0) > 0 ? (H - row0) : 0) : part_h;
                const int ha = rows < 8 ? rows : 8, hb = rows > 8 ? rows - 8 : 0;   // rows [0, ha) and [hb, rows)
                const int crow = row0 + (rr == 0 ? 0 : rr == 1 ? hb : rr == 2 ? ha : rows);
                const int strip = chunk0 + k;
                const int xcol = strip * (TILE_CX - 1) < W ? strip * (TILE_CX - 1) : W;
                float dv[3], th;
                uint32_t path;
                make_ray(cam, (float)xcol, (float)(H - crow), dv);
                uint32_t key = ctx.template trace<true>(dv, th, path);
                const uint32_t kind = key & 7u;
                const bool ok = (kind == KIND_FLOOR || kind == KIND_CEIL || kind == KIND_WALL) && !(path & 0x80000000u);
                const uint32_t path_raw = path;   // the crossings made, whatever was met in the end
                if (!ok) { key = 0xF0000000u | (uint32_t)lane; path = (uint32_t)lane; }   // equal to no other lane's
                // lane (rr = 0, k) owns item (strip, q): its corners sit at lanes +0 +1 (top), +8 +9 (rows - 8), +16 +17 (8), +24 +25 (bottom)
                uint32_t nk[8], np[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int src = (lane + (j >> 1) * 8 + (j & 1)) & (WAVE - 1);
                    nk[j] = __shfl(key, src); np[j] = __shfl(path, src);
                }
                auto same = [&](int a, int b) { return nk[a] == nk[b] && np[a] == np[b]; };
                const bool top_u = ok && same(0, 1) && same(0, 4) && same(0, 5);      // corner rows 0 and ha
                const bool bot_u = same(2, 3) && same(2, 6) && same(2, 7);   // corner rows hb and rows (a lane that is not ok equals nobody)
                const bool all_u = top_u && bot_u && same(0, 2);
                int ur0 = 0, ur1 = 0;   // uniform rows [ur0, ur1) of the item
                uint32_t ukey = 0;
                if (all_u) { ur1 = rows; ukey = nk[0]; }
                else if (top_u && ha < rows) { ur1 = ha; ukey = nk[0]; }
                else if (bot_u && hb > 0) { ur0 = hb; ur1 = rows; ukey = nk[2]; }
                // The leading portal crossings the item's four outer corner rays share are made by every ray of the item
                // (same convexity argument): its corner passes start in the room behind them (found by walking this
                // lane's own ray, the top-left corner) and the 8-sample pixels add the count to their own.
                int common = common_crossings(path_raw, __shfl(path_raw, (lane + 1) & (WAVE - 1)));
                const int c1 = common_crossings(path_raw, __shfl(path_raw, (lane + 24) & (WAVE - 1)));
                const int c2 = common_crossings(path_raw, __shfl(path_raw, (lane + 25) & (WAVE - 1)));
                common = common < c1 ? common : c1; common = common < c2 ? common : c2;
                if ((d.debug_flags & 8) || ctx.cam_room < 0) common = 0;
                int item_room = ctx.cam_room;
                if (common > 0) {
                    const int r = POLY ? walk_rooms_poly(rooms, ctx.cam_room, cam.eye, dv, common) : walk_rooms(rooms, ctx.cam_room, cam.eye, dv, common);
                    if (r >= 0) item_room = r; else common = 0;
                }
                // Which boxes can touch this item at all?  The item's rays lie in the cone around its centre ray that holds its
                // corner rays (half angle a); box b's pixel-inflated bounding sphere (the one box_may_touch tests) subtends
                // asin(R / dist) around the direction to its centre: no ray of the item meets it unless the two directions are
                // within a + b of each other.  Conservative (2 % on R, 0.02 on the cosine); evaluated once per item.
                uint32_t boxes = 0xFFFFFFFFu;
                if (NBOX > 1 && ctx.boxes_in_view) {
                    boxes = 0;
                    const int ix0 = strip * (TILE_CX - 1), iwi = (W - ix0) < (TILE_CX - 1) ? (W - ix0) : (TILE_CX - 1);
                    float c[3], cn;
                    make_ray(cam, (float)ix0 + 0.5f * (float)iwi, (float)(H - row0) - 0.5f * (float)rows, c);
                    cn = __builtin_amdgcn_rsqf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                    float cos_a = 1.0f;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float e[3];
                        make_ray(cam, (float)(ix0 + ((j & 1) ? iwi : 0)), (float)(H - row0 - ((j & 2) ? rows : 0)), e);
                        const float en = __builtin_amdgcn_rsqf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                        const float dt = (c[0] * e[0] + c[1] * e[1] + c[2] * e[2]) * cn * en;
                        cos_a = dt < cos_a ? dt : cos_a;
                    }
                    cos_a = cos_a - 0.002f;   // the corner rays' cone, a little wider
                    cos_a = cos_a < 0.0f ? 0.0f : cos_a;
                    const float sin_a = __builtin_sqrtf(1.0f - cos_a * cos_a);
#pragma unroll 1
                    for (int bi = 0; bi < NBOX; bi++) {
                        const float *fb_ = fc + bi * FC_BOX_STRIDE;
                        const float ox = fb_[FC_CULL_OC], oy = fb_[FC_CULL_OC + 1], oz = fb_[FC_CULL_OC + 2], ccp = fb_[FC_CULL_CC_PIXEL];
                        const float oc2 = ox * ox + oy * oy + oz * oz;
                        bool touch = true;
                        if (ccp > 0.0f) {   // the eye is outside the inflated sphere
                            const float idist = __builtin_amdgcn_rsqf(oc2);
                            float sin_b = 1.02f * __builtin_sqrtf(fmaxf(oc2 - ccp, 0.0f)) * idist;
                            sin_b = sin_b > 1.0f ? 1.0f : sin_b;
                            const float cos_b = __builtin_sqrtf(1.0f - sin_b * sin_b);
                            const float cos_t = (c[0] * ox + c[1] * oy + c[2] * oz) * cn * idist;
                            touch = cos_t >= cos_a * cos_b - sin_a * sin_b - 0.02f;
                        }
                        boxes |= touch ? 1u << bi : 0u;
                    }
                }
                if (rr == 0 && k < 7 && strip < n_strips)
                    item_res[strip * 4 + q] = make_uint4(ukey, (uint32_t)(ur0 | (ur1 << 16)), boxes, (uint32_t)((item_room & 0xFFFF) | (common << 16)));
            }
        }
        __syncthreads();
    }

    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(cam_room_s + 1, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        if (part >= 0) item = split_x ? item + part * half_strips * 4 : ((item >> 1) << 2) + (item & 1) + 2 * part;
        const int x0 = (item >> 2) * (TILE_CX - 1), row0 = (item & 3) * part_h;
        const int rows = (H - row0) < part_h ? (H - row0) : part_h;   // pixel rows of this item
        if (rows <= 0) continue;
        int prow0 = row0, prows = rows;   // the rows left for the per-pixel corner passes
        int item_skip = 0, item_room = ctx.cam_room;
        uint32_t item_boxes = 0xFFFFFFFFu;   // the boxes whose pixel-inflated sphere can meet the item's rays
        if (d.debug_flags & 128) continue;   // experiment: prologue + copy-out only
        if (!(d.debug_flags & (1 | 32))) {
            const int wi = (W - x0) < (TILE_CX - 1) ? (W - x0) : (TILE_CX - 1);   // pixel columns of this item
            const uint4 res = item_res[item];   // the frame-level pre-test's verdict (workgroup-uniform address)
            const uint32_t ukey = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.x);
            const int urw = __builtin_amdgcn_readfirstlane((int)res.y);
            const int ur0 = urw & 0xFFFF, ur1 = urw >> 16;   // uniform rows [ur0, ur1) of the item
            if (NBOX > 1) item_boxes = (uint32_t)__builtin_amdgcn_readfirstlane((int)res.z);
            const int rw = __builtin_amdgcn_readfirstlane((int)res.w);
            item_skip = rw >> 16;                                        // crossings every ray of the item makes, and the room behind them
            item_room = (rw & 0xFFFF) == 0xFFFF ? -1 : (rw & 0xFFFF);
            if (ur1 > ur0) {
                if (ur0 == 0 && ur1 == rows) prows = 0;
                else if (ur0 == 0) { prow0 = row0 + ur1; prows = rows - ur1; }
                else prows = ur0;
            }
            const int col = lane & (TILE_CX - 1);
            for (int r0 = ur0; r0 < ur1; r0 += TILE_CY) {
                const int px = x0 + col, py = row0 + r0 + lane / TILE_CX;
                const bool is_pixel = col < wi && py < row0 + ur1;
                bool interior = is_pixel;
                if (interior && ctx.boxes_in_view && box_may_touch(px, py, item_boxes)) interior = false;
                emit(px, py, is_pixel, interior, ukey, 0, item_boxes);
            }
            if (prows <= 0) continue;
        }
        if (d.debug_flags & 64) continue;   // experiment: no per-pixel corner passes (prologue + pre-test + copy-out only)
        const int n_pass_i = (prows + 1 + TILE_CY - 1) / TILE_CY;   // corner rows of the remaining rows: prows + 1
        uint32_t prev_key = 0, prev_path = 0;
        for (int p = 0; p < n_pass_i; p++) {
            const int crow = prow0 + p * TILE_CY + cj;        // corner row (0 .. H), window y (up) = H - crow
            float dv[3], th;
            uint32_t path;
            make_ray(cam, (float)(x0 + ci), (float)(H - crow), dv);
            uint32_t key = ctx.template trace_from<true>(item_room, dv, th, path);
            const uint32_t kind = key & 7u;
            bool ok = (kind == KIND_FLOOR || kind == KIND_CEIL || kind == KIND_WALL) && !(path & 0x80000000u);
            // the pixel whose bottom-left corner this lane traced: corners (crow-1, ci), (crow-1, ci+1),
            // (crow, ci), (crow, ci+1); row crow-1 comes from the lanes above or from the previous pass
            // right neighbour: a shift inside the 16-lane row (DPP, no LDS traffic; lane 15 of a row is no pixel)
            const uint32_t k_br = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
            const uint32_t p_br = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)path, 0x101, 0xf, 0xf, false);
            // row above: lanes 16..63 want this pass' lane - 16, lanes 0..15 the previous pass' last row (lane + 48):
            // both are "lane - 16 modulo 64" of a register that holds the previous pass' values in its last row
            const uint32_t ck = cj == TILE_CY - 1 ? prev_key : key, cp = cj == TILE_CY - 1 ? prev_path : path;
            const uint32_t k_tl = __shfl(ck, (lane - TILE_CX) & (WAVE - 1)), k_tr = __shfl(ck, (lane - TILE_CX + 1) & (WAVE - 1));
            const uint32_t p_tl = __shfl(cp, (lane - TILE_CX) & (WAVE - 1)), p_tr = __shfl(cp, (lane - TILE_CX + 1) & (WAVE - 1));
            prev_key = key; prev_path = path;
            const int px = x0 + ci, py = crow - 1;
            const bool is_pixel = ci < TILE_CX - 1 && px < W && py >= prow0 && py < prow0 + prows && !(p == 0 && cj == 0);
            bool interior = !(d.debug_flags & 1) && is_pixel && ok && key == k_br && key == k_tl && key == k_tr &&
                            path == p_br && path == p_tl && path == p_tr;
            if (interior && ctx.boxes_in_view && box_may_touch(px, py, item_boxes)) interior = false;
            int skip = common_crossings4(path, p_br, p_tl, p_tr) + item_skip;
            skip = skip < skip_max ? skip : skip_max;
            emit(px, py, is_pixel, interior, key, skip, item_boxes);
        }
    }
    // Leftovers (< 64 per wave and queue): pooled over the workgroup and dealt out again in full batches - interior
    // pixels to the waves from the first up, 8-sample pixels from the last down - instead of every wave
    // running two partly filled batches.
    int *left = cam_room_s + 4;
    if (lane == 0) { left[wave] = iq_count; left[n_waves + wave] = q_count; }
    if constexpr (NBOX > MWB_MAX_BOXES) { if (lane == 0) mleft[wave] = mq_count; }
    __syncthreads();
    if (!(d.debug_flags & 4)) {
        int g = tid, owner = -1;
#pragma unroll
        for (int w = 0; w < THREADS / WAVE; w++) {
            const int c = left[w];
            if (owner < 0) { if (g < c) owner = w; else g -= c; }
        }
        if (owner >= 0) {
            const int p = ipix[owner * QUEUE_CAP + g];
            ctx.pixel_interior(p & wmask, p >> wshift, ikeys[owner * QUEUE_CAP + g]);
        }
    }
    if (!(d.debug_flags & 2)) {
        int g = THREADS - 1 - tid, owner = -1;
#pragma unroll
        for (int w = 0; w < THREADS / WAVE; w++) {
            const int c = left[n_waves + w];
            if (owner < 0) { if (g < c) owner = w; else g -= c; }
        }
        if (owner >= 0) {
            const int q = queues[owner * QUEUE_CAP + g];
            ctx.pixel_full(q & wmask, (q >> wshift) & hmask, q >> qshift);
        }
    }
    if constexpr (NBOX > MWB_MAX_BOXES) {
        if (!(d.debug_flags & 2)) {   // the mesh pixels, pooled the same way: 64-pixel batches dealt out to the waves in turn
            int total = 0;
#pragma unroll
            for (int w = 0; w < THREADS / WAVE; w++) total += mleft[w];
            for (int base = 0; base < total; base += THREADS) {
                int g = base + tid, owner = -1;
#pragma unroll
                for (int w = 0; w < THREADS / WAVE; w++) {
                    const int c = mleft[w];
                    if (owner < 0) { if (g < c) owner = w; else g -= c; }
                }
                if (base + (tid & ~(WAVE - 1)) < total) {   // this wave has pixels to do (wave-uniform: pixels_mesh is a wave's joint work)
                    const int q = owner >= 0 ? mqueues[owner * MQ_CAP + g] : 0;
                    ctx.pixels_mesh(q & wmask, (q >> wshift) & hmask, q >> qshift, owner >= 0);
                }
            }
        }
    }
    __syncthreads();
    {   // framebuffer LDS -> HBM, 16 bytes per lane where the alignment allows
        const int nbytes = W * H * 3;
        uint8_t *dst = d.obs + (size_t)e * nbytes;
        if (TILED) {
            if (d.layout == MWB_LAYOUT_HWC) {   // the tile's rows into the big frame
                uint8_t *big = d.obs + ((size_t)e * d.W * d.H + (size_t)ty0 * d.W + tx0) * 3;
                for (int i = tid; i < H * W * 3; i += THREADS) {
                    const int row = i / (W * 3), col = i - row * (W * 3);
                    big[(size_t)row * d.W * 3 + col] = fb[i];
                }
            } else {   // CWH: the tile is [3][W][H] in LDS, its columns go into the planes [3][d.W][d.H]
                uint8_t *big = d.obs + (size_t)e * d.W * d.H * 3;
                for (int i = tid; i < H * W * 3; i += THREADS) {
                    const int q = i / (W * H), r = i - q * (W * H), x = r / H, y = r - x * H;
                    big[((size_t)q * d.W + tx0 + x) * d.H + ty0 + y] = fb[i];
                }
            }
        } else if (part < 0) {
            copy_frame_range<THREADS>(dst, fb, 0, nbytes, tid);
        } else if (!split_x) {   // HWC: rows [r0, r1)
            const int mid = 2 * part_h < H ? 2 * part_h : H;
            const int r0 = part ? mid : 0, r1 = part ? H : mid;
            copy_frame_range<THREADS>(dst, fb, r0 * W * 3, r1 * W * 3, tid);
        } else {                 // CWH: columns [xa, xb) of each channel plane
            const int midx = half_strips * (TILE_CX - 1) < W ? half_strips * (TILE_CX - 1) : W;
            const int xa = part ? midx : 0, xb = part ? W : midx;
            for (int q = 0; q < 3; q++) copy_frame_range<THREADS>(dst, fb, (q * W + xa) * H, (q * W + xb) * H, tid);
        }
        // Fused frame stack (mwb_stack_enable with MWB_STACK_FUSED; CWH frames): the new frame also goes straight into the
        // newest three planes of the env's sliding window - u8 -> f32 on the way out of LDS - and an env that was
        // regenerated in this pass gets its history planes zeroed (VecPyTorchFrameStack, envs.py:149-156): no stack pass.
        if (d.stk && d.layout == MWB_LAYOUT_CWH) {
            const int plane = W * H, C = d.stk_C;
            const size_t env_base = ((size_t)e * d.stk_K + d.stk_pos) * plane;   // first element of the window
            int xa = 0, xb = W;
            if (part >= 0) { const int midx = half_strips * (TILE_CX - 1) < W ? half_strips * (TILE_CX - 1) : W; xa = part ? midx : 0; xb = part ? W : midx; }
            if (LOOPED || d.reset_set[e]) {   // regenerated env (side-stream list, or a pass that regenerates in place; never the bulk pass)
                const int n4 = (C - 3) * plane / 4;   // a half-frame workgroup zeroes half of the history
                const int i0 = part < 0 ? 0 : (part ? n4 / 2 : 0), i1 = part < 0 ? n4 : (part ? n4 : n4 / 2);
                if (d.stk_float) { float4 *z = (float4 *)((float *)d.stk + env_base); for (int i = i0 + tid; i < i1; i += THREADS) z[i] = make_float4(0, 0, 0, 0); }
                else { uint32_t *z = (uint32_t *)((uint8_t *)d.stk + env_base); for (int i = i0 + tid; i < i1; i += THREADS) z[i] = 0u; }
            }
            for (int q = 0; q < 3; q++) {
                const int b0 = (q * W + xa) * H, b1 = (q * W + xb) * H;   // byte range of this channel plane's columns in fb (multiples of 4: W*H % 4 == 0 checked, H*15 ... see host check)
                const size_t dst0 = env_base + (size_t)(C - 3) * plane;
                if (d.stk_float) {
                    float *o = (float *)d.stk + dst0;
                    if (((b0 | b1) & 3) == 0) {
                        for (int i = b0 / 4 + tid; i < b1 / 4; i += THREADS) {
                            const uint32_t p = ((const uint32_t *)fb)[i];
                            ((float4 *)o)[i] = make_float4((float)(p & 255u), (float)((p >> 8) & 255u), (float)((p >> 16) & 255u), (float)(p >> 24));
                        }
                    } else {
                        for (int i = b0 + tid; i < b1; i += THREADS) o[i] = (float)fb[i];
                    }
                } else {
                    copy_frame_range<THREADS>((uint8_t *)d.stk + dst0, fb, b0, b1, tid);
                }
            }
        }
    }
}

// MODE 0: every env; 1: only the envs regenerated this step (side stream, through the compact list);
// 2: all the others (bulk).  A template parameter so that the three launches carry distinct kernel names
// in profiles.
// NBOX = MWB_MAX_ENTS: the entity tasks' instantiation (mesh BVH walks, frames): 4 workgroups per CU (128 VGPRs) instead of 5
template <int THREADS, int MODE, int NBOX, bool POLY = false>
__global__ void __launch_bounds__(THREADS, NBOX > MWB_MAX_BOXES ? ENT_WGS_PER_CU : 5) render_kernel(MwbDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (MODE == 1) {
        __builtin_amdgcn_s_setprio(3);   // the few regenerated envs, beside the bulk render (see reset_kernel)
        const int count = d.reset_count[0];
        for (int li = blockIdx.x; li < count; li += gridDim.x) {
            render_env<THREADS, NBOX, true, POLY>(d, d.reset_list[li], -1, smem);
            __syncthreads();   // LDS is reused by the next env of this block
        }
    } else {
        // Workgroups take the envs in the order of decreasing cost measured by the previous launch (frames change
        // little from step to step): a frame's time varies from 0.5x to 2.5x the mean, and starting the slow ones
        // first keeps the launch from ending on a few stragglers.  The last d.split_envs of that order - the
        // cheapest frames - run as two half-frame workgroups each.
        const int whole = d.N - d.split_envs;
        const int b = blockIdx.x;
        const int slot = b < whole ? b : whole + ((b - whole) >> 1);
        const int part = b < whole ? -1 : ((b - whole) & 1);
        const int e = __builtin_amdgcn_readfirstlane(d.order_bufs[d.order_state[0] & 1][slot]);
        if (MODE == 2 && d.reset_set[e]) return;   // block-uniform
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // read by every wave: stays in scalar registers
        render_env<THREADS, NBOX, false, POLY>(d, e, part, smem);
        if (threadIdx.x == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            if (d.wg_ts) { d.wg_ts[2 * b] = t0; d.wg_ts[2 * b + 1] = t1; }
            const uint32_t dt = (uint32_t)(t1 - t0);
            if (part < 0) { d.cost[2 * e] = dt; d.cost[2 * e + 1] = 0; }
            else d.cost[2 * e + part] = dt;
        }
    }
}

// The agent's view at any size (mwb_render_view): a grid of tiles per env, one workgroup each.  d.W x d.H is the view's size, d.obs /
// d.depth / d.frame the caller's frame, the optional depth map and the frame constants prepared for that size.
// MODE 0: every env; 1: the regenerated envs (compact list, grid-stride); 2: all the others.  The entity tasks render EVERY frame
// this way (MWB_TILE, default 40 x 30: four workgroups per frame): a close-up of a 5 000-triangle ball makes one frame cost fifty
// times the average, and a launch is as slow as its slowest workgroup (measured: 0.7 waves per SIMD resident on average).
#define VIEW_TILE_W 75   // mwb_render_view: 5 strips of 15 pixels
#define VIEW_TILE_H 60
template <int THREADS, int MODE, int NBOX, bool POLY>
__global__ void __launch_bounds__(THREADS, NBOX > MWB_MAX_BOXES ? ENT_WGS_PER_CU : 5) render_view_kernel(MwbDev d, int tiles_x, int tiles_y, int tile_w, int tile_h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int per_env = tiles_x * tiles_y;
    if (MODE == 1) {
        __builtin_amdgcn_s_setprio(3);
        const int total = d.reset_count[0] * per_env;
        for (int w = blockIdx.x; w < total; w += gridDim.x) {
            const int li = w / per_env, t = w - li * per_env;
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            const int tx0 = tx * tile_w, ty0 = ty * tile_h;
            render_env<THREADS, NBOX, true, POLY, true>(d, d.reset_list[li], -1, smem, tx0, ty0, d.W - tx0 < tile_w ? d.W - tx0 : tile_w, d.H - ty0 < tile_h ? d.H - ty0 : tile_h);
            __syncthreads();
        }
        return;
    }
    const int e = blockIdx.x / per_env, t = blockIdx.x - e * per_env;
    if (MODE == 2 && d.reset_set[e]) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int tx0 = tx * tile_w, ty0 = ty * tile_h;
    const int tw = d.W - tx0 < tile_w ? d.W - tx0 : tile_w, th = d.H - ty0 < tile_h ? d.H - ty0 : tile_h;
    render_env<THREADS, NBOX, false, POLY, true>(d, e, -1, smem, tx0, ty0, tw, th);
}

// ================================================================================== frame stack
// VecPyTorchFrameStack.step_wait / reset (pytorch-a2c-ppo-acktr/envs.py:149-162) fused with VecPyTorch's
// uint8 -> float conversion (envs.py:128): one streaming pass, 16 bytes per lane.  HBM-bound: per env it
// reads (C-3) planes + the new observation and writes C planes.
template <typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(256) stack_kernel(MwbDev d, V *__restrict__ stack, int C, int after_reset) {
    const int e = blockIdx.y;
    const int plane4 = (d.W * d.H) / 4;   // vectors of 4 pixels per channel plane
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= plane4) return;
    V *base = stack + (size_t)e * C * plane4;
    const bool clear = after_reset || d.done[e];
    V zero;
    memset(&zero, 0, sizeof(V));
    for (int c = 0; c < C - 3; c++) {
        V v = zero;
        if (!clear) v = base[(size_t)(c + 3) * plane4 + j];
        base[(size_t)c * plane4 + j] = v;
    }
    const uint32_t *obs = (const uint32_t *)(d.obs + (size_t)e * 3 * d.W * d.H);   // CWH: [3][W][H]
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t p = obs[(size_t)k * plane4 + j];
        V v;
        if constexpr (IS_FLOAT) {
            v.x = (float)(p & 255u); v.y = (float)((p >> 8) & 255u); v.z = (float)((p >> 16) & 255u); v.w = (float)(p >> 24);
        } else {
            v = p;
        }
        base[(size_t)(C - 3 + k) * plane4 + j] = v;
    }
}

// The same stack as a SLIDING WINDOW over K > C channel planes per env: the view of step t is planes [pos, pos + C); a step
// moves the window three planes on and writes only the new frame (3 planes) - the C - 3 planes of history stay where they
// are - and zeroes the history of the envs whose episode ended; when the window reaches the end of the K planes the
// history is copied back to the front (the classic pass, once every (K - C) / 3 + 1 steps).  mode: 0 slide to `pos` (the
// window's new first plane), 1 wrap (history from `from`, window at 0), 2 after reset (window at 0, history zeroed).
template <typename V, bool IS_FLOAT>
__global__ void __launch_bounds__(256) stack_slide_kernel(MwbDev d, V *__restrict__ stack, int C, int K, int pos, int from, int mode) {
    const int e = blockIdx.y;
    const int plane4 = (d.W * d.H) / 4;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= plane4) return;
    V *base = stack + (size_t)e * K * plane4;
    const bool clear = mode == 2 || (mode < 3 && d.done[e]);
    V zero;
    memset(&zero, 0, sizeof(V));
    if (mode == 3) {   // fused stack: the history back to the front, nothing else (the render kernels write frames and zero finished envs)
        for (int c = 0; c < C - 3; c++) base[(size_t)c * plane4 + j] = base[(size_t)(from + 3 + c) * plane4 + j];
        return;
    }
    if (mode == 1 && !clear) {
        for (int c = 0; c < C - 3; c++) base[(size_t)c * plane4 + j] = base[(size_t)(from + 3 + c) * plane4 + j];
    } else if (clear) {
        for (int c = 0; c < C - 3; c++) base[(size_t)(pos + c) * plane4 + j] = zero;
    }
    const uint32_t *obs = (const uint32_t *)(d.obs + (size_t)e * 3 * d.W * d.H);   // CWH: [3][W][H]
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t p = obs[(size_t)k * plane4 + j];
        V v;
        if constexpr (IS_FLOAT) {
            v.x = (float)(p & 255u); v.y = (float)((p >> 8) & 255u); v.z = (float)((p >> 16) & 255u); v.w = (float)(p >> 24);
        } else {
            v = p;
        }
        base[(size_t)(pos + C - 3 + k) * plane4 + j] = v;
    }
}
void mwb_launch_stack_slide(const MwbDev &d, void *stack, int nstack, int planes, int dtype, int pos, int from, int mode, hipStream_t s) {
    const int plane4 = (d.W * d.H) / 4;
    dim3 grid((plane4 + 255) / 256, d.N);
    if (dtype == 1) stack_slide_kernel<float4, true><<<grid, dim3(256), 0, s>>>(d, (float4 *)stack, nstack * 3, planes, pos, from, mode);
    else stack_slide_kernel<uint32_t, false><<<grid, dim3(256), 0, s>>>(d, (uint32_t *)stack, nstack * 3, planes, pos, from, mode);
}

void mwb_launch_stack(const MwbDev &d, void *stack, int nstack, int dtype, int after_reset, hipStream_t s) {
    const int plane4 = (d.W * d.H) / 4;
    dim3 grid((plane4 + 255) / 256, d.N);
    if (dtype == 1) stack_kernel<float4, true><<<grid, dim3(256), 0, s>>>(d, (float4 *)stack, nstack * 3, after_reset);
    else stack_kernel<uint32_t, false><<<grid, dim3(256), 0, s>>>(d, (uint32_t *)stack, nstack * 3, after_reset);
}

// ============================================================================== small utilities
__global__ void intersect_kernel(MwbDev d, int e, int ent, double x, double z, double radius, int *result) {
    // MiniWorldEnv.intersect(ent, pos, radius), miniworld.py:933-959: walls, then the entities in list order
    // (boxes, agent) except `ent` itself
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int ns = d.n_segs[e], res = 0;
    for (int i = 0; i < ns && !res; i++) {
        double q[4];
        for (int c = 0; c < 4; c++) q[c] = d.segs[(size_t)(i * 4 + c) * d.N + e];
        if (seg_hit(q, x, z, radius)) res = 1;
    }
    if (d.ent_task) {   // the entity list in its current order, radii from the entities (tagged sums: the query radius is a Python float)
        const uint8_t *ord = d.ent_order + (size_t)e * MWB_ORDER_STRIDE;
        const int n_ord = d.n_order[e];
        for (int k = 0; k < n_ord && !res; k++) {
            const int slot = ord[k];
            if (slot == (ent == d.n_boxes ? MWB_ENT_AGENT : ent)) continue;
            double ox, oz, orad; bool of32 = false;
            if (slot == MWB_ENT_AGENT) { ox = d.agent_x[e]; oz = d.agent_z[e]; orad = d.agent_radius; }
            else { const size_t be = (size_t)slot * d.N + e; ox = d.box_x[be]; oz = d.box_z[be]; orad = d.ent_radius[be]; of32 = MWB_META_RADF32(d.ent_meta[be]) != 0; }
            const double ddx = ox - x, ddz = oz - z;
            if (sqrt(ddx * ddx + 0.0 + ddz * ddz) < tagged_add(radius, false, orad, of32)) res = 2 + (slot == MWB_ENT_AGENT ? d.n_boxes : slot);
        }
        *result = res;
        return;
    }
    for (int b = 0; b < d.n_boxes && !res; b++) {
        if (b == ent) continue;
        double ddx = d.box_x[(size_t)b * d.N + e] - x, ddz = d.box_z[(size_t)b * d.N + e] - z;
        if (sqrt(ddx * ddx + 0.0 + ddz * ddz) < radius + box_radius(d.box_size[(size_t)b * d.N + e])) res = 2 + b;
    }
    if (!res && ent != d.n_boxes) {   // the agent is the last entity of the list
        double ddx = d.agent_x[e] - x, ddz = d.agent_z[e] - z;
        if (sqrt(ddx * ddx + 0.0 + ddz * ddz) < radius + d.agent_radius) res = 2 + d.n_boxes;
    }
    *result = res;
}

// ====================================================================================== launch
#define RENDER_THREADS 256
size_t mwb_reset_lds_bytes(const MwbDev &d) {
    size_t b = (size_t)d.R_max * sizeof(WRoom) + (size_t)d.R_max * sizeof(double) + (size_t)((d.R_max + 3) & ~3) * sizeof(int) + 624 * sizeof(uint32_t);
    if (d.ent_task) b += (size_t)MWB_MAX_ENTS * (11 * sizeof(double) + 3 * sizeof(int)) + 8 * sizeof(int);   // the entity tasks' slot records
    if (d.task == MWB_TASK_MAZE) b += (size_t)(d.R_max + 1) / 2 * 13 + 16;   // rows * cols cells: 3 ints + 1 flag each, one frame more
    return (b + 15) & ~(size_t)15;
}
static size_t render_lds_bytes_for(const MwbDev &d, int W, int H) {
    size_t b = (((size_t)d.R_max * d.room_words * 4 + 15) & ~(size_t)15) + (size_t)d.frame_words * 4 + sizeof(TexLds) * d.n_tex + 16 + 2 * (RENDER_THREADS / WAVE) * sizeof(int) +
               (RENDER_THREADS / WAVE) * QUEUE_CAP * (2 * sizeof(uint16_t) + sizeof(uint32_t)) + (size_t)ITEM_RES_BYTES(W) + (size_t)W * H * 3;
    b = (b + 15) & ~(size_t)15;
    if (d.ent_task) b += (RENDER_THREADS / WAVE) * MQ_CAP * sizeof(uint16_t) + 16 + 16 * MWB_NUM_MESHES + (RENDER_THREADS / WAVE) * MB_WAVE_BYTES;   // the mesh-pixel queues, their leftover counts, the mesh descriptors, the batch scratch
    return (b + 15) & ~(size_t)15;
}
// d: the handle's MwbDev with W / H / obs / depth / frame / want_depth / layout set for the view
#define LIST_GRID 1280   // blocks that walk the compact list of regenerated envs: a handful per step - but ALL of them in the step at
                        // which a whole batch hits the episode limit together (a block with nothing to do exits at once)
template <int MODE>
static int launch_tiles(const MwbDev &d, int tile_w, int tile_h, hipStream_t s) {
    const int tiles_x = (d.W + tile_w - 1) / tile_w, tiles_y = (d.H + tile_h - 1) / tile_h;
    const size_t lds = render_lds_bytes_for(d, tile_w, tile_h);
    if (lds > 160 * 1024) return -1;
    const size_t blocks = MODE == 1 ? (size_t)(d.N < LIST_GRID ? d.N : LIST_GRID) * tiles_x * tiles_y : (size_t)d.N * tiles_x * tiles_y;
    const dim3 g((unsigned)blocks), b(RENDER_THREADS);
#define RV(NB, PL)                                                                                                                 \
    do {                                                                                                                           \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)render_view_kernel<RENDER_THREADS, MODE, NB, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2; \
        render_view_kernel<RENDER_THREADS, MODE, NB, PL><<<g, b, lds, s>>>(d, tiles_x, tiles_y, tile_w, tile_h);                   \
    } while (0)
    if (d.ent_task) RV(MWB_MAX_ENTS, false);
    else if (d.poly) RV(1, true);
    else if (d.n_boxes == 6) RV(6, false);
    else if (d.n_boxes == 2) RV(2, false);
    else RV(1, false);
#undef RV
    return 0;
}
int mwb_launch_render_view(const MwbDev &d, hipStream_t s) { return launch_tiles<0>(d, VIEW_TILE_W, VIEW_TILE_H, s); }
void mwb_view_tile(int *w, int *h) { *w = VIEW_TILE_W; *h = VIEW_TILE_H; }

size_t mwb_render_lds_bytes(const MwbDev &d) {
    if (d.tile_w > 0) return render_lds_bytes_for(d, d.tile_w, d.tile_h);   // observations rendered in tiles (large frames)
    if (d.ent_task) return render_lds_bytes_for(d, d.W, d.H);
    size_t b = (((size_t)d.R_max * d.room_words * 4 + 15) & ~(size_t)15) + (size_t)d.frame_words * 4 + sizeof(TexLds) * d.n_tex + 16 + 2 * (RENDER_THREADS / WAVE) * sizeof(int) +
               (RENDER_THREADS / WAVE) * QUEUE_CAP * (2 * sizeof(uint16_t) + sizeof(uint32_t)) + (size_t)ITEM_RES_BYTES(d.W) + (size_t)d.W * d.H * 3;
    b += (size_t)(d.debug_flags >> 8) * 128;   // MWB_DEBUG bits 8+: units of 128 B of LDS padding (occupancy experiments)
    return (b + 15) & ~(size_t)15;
}

void mwb_launch_step(const MwbDev &d, const int32_t *actions, const uint8_t *skip, hipStream_t s) {
    if (d.ent_task) { hipLaunchKernelGGL(step_ents_kernel, dim3((d.N + 63) / 64), dim3(64), 0, s, d, actions, skip); return; }
    const dim3 g((d.N + 63) / 64), b(64 * STEP_PARTS);
    if (d.n_boxes == 1) hipLaunchKernelGGL(step_kernel<1>, g, b, 0, s, d, actions, skip);
    else if (d.n_boxes == 2) hipLaunchKernelGGL(step_kernel<2>, g, b, 0, s, d, actions, skip);
    else hipLaunchKernelGGL(step_kernel<MWB_MAX_BOXES>, g, b, 0, s, d, actions, skip);
}
__global__ void clear_list_kernel(MwbDev d) { d.reset_count[0] = 0; }

void mwb_launch_clear_list(const MwbDev &d, hipStream_t s) { hipLaunchKernelGGL(clear_list_kernel, dim3(1), dim3(1), 0, s, d); }
void mwb_launch_mark_reset(const MwbDev &d, const uint8_t *mask, hipStream_t s) {
    hipLaunchKernelGGL(mark_reset_kernel, dim3((d.N + 255) / 256), dim3(256), 0, s, d, mask);
}
typedef void (*reset_fn_t)(MwbDev);
static reset_fn_t reset_fn(int task);
int mwb_prepare_kernels(const MwbDev &d) {
    // opt in to more than the default 64 KB of dynamic LDS where a large world needs it (160 KB per CU)
    size_t r = mwb_reset_lds_bytes(d), q = mwb_render_lds_bytes(d);
    if (r > 160 * 1024 || q > 160 * 1024) return -1;
    {   // the render kernel's pixel queues hold (py << ceil(log2 W)) | px in 16 bits
        int wshift = 0;
        const int qw = d.tile_w > 0 ? d.tile_w : d.W, qh = d.tile_w > 0 ? d.tile_h : d.H;
        while ((1 << wshift) < qw) wshift++;
        if (((size_t)qh << wshift) > 65536) return -3;
    }
    if (r > 64 * 1024 && hipFuncSetAttribute((const void *)reset_fn(d.task), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r) != hipSuccess) return -2;
    if (q > 64 * 1024) {   // large mazes, or any task at a large observation size (the W*H*3 frame is in LDS too)
        const void *fns[3][3] = {{(const void *)render_kernel<RENDER_THREADS, 0, 1>, (const void *)render_kernel<RENDER_THREADS, 1, 1>, (const void *)render_kernel<RENDER_THREADS, 2, 1>},
                                 {(const void *)render_kernel<RENDER_THREADS, 0, 2>, (const void *)render_kernel<RENDER_THREADS, 1, 2>, (const void *)render_kernel<RENDER_THREADS, 2, 2>},
                                 {(const void *)render_kernel<RENDER_THREADS, 0, 6>, (const void *)render_kernel<RENDER_THREADS, 1, 6>, (const void *)render_kernel<RENDER_THREADS, 2, 6>}};
        const void *pfns[3] = {(const void *)render_kernel<RENDER_THREADS, 0, 1, true>, (const void *)render_kernel<RENDER_THREADS, 1, 1, true>, (const void *)render_kernel<RENDER_THREADS, 2, 1, true>};
        const void *efns[3] = {(const void *)render_kernel<RENDER_THREADS, 0, MWB_MAX_ENTS>, (const void *)render_kernel<RENDER_THREADS, 1, MWB_MAX_ENTS>, (const void *)render_kernel<RENDER_THREADS, 2, MWB_MAX_ENTS>};
        for (int m = 0; m < 3; m++)
            if (hipFuncSetAttribute(d.ent_task ? efns[m] : d.poly ? pfns[m] : fns[d.n_boxes == 6 ? 2 : d.n_boxes == 2 ? 1 : 0][m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)q) != hipSuccess) return -2;
    }
    return 0;
}

static reset_fn_t reset_fn(int task) {
    switch (task) {
#define RK(t) case t: return reset_kernel<t>;
    RK(MWB_TASK_HALLWAY) RK(MWB_TASK_ONEROOM) RK(MWB_TASK_FOURROOMS) RK(MWB_TASK_MAZE) RK(MWB_TASK_TMAZE) RK(MWB_TASK_TMAZE_TWOBOX)
    RK(MWB_TASK_SIM2REAL_GOTO) RK(MWB_TASK_SIM2REAL_PUSH) RK(MWB_TASK_PUTNEXT) RK(MWB_TASK_YMAZE) RK(MWB_TASK_PICKUPOBJS)
    RK(MWB_TASK_ROOMOBJS) RK(MWB_TASK_COLLECTHEALTH) RK(MWB_TASK_THREEROOMS) RK(MWB_TASK_SIGN) RK(MWB_TASK_SIDEWALK) RK(MWB_TASK_WALLGAP)
#undef RK
    }
    return reset_kernel<MWB_TASK_MAZE>;
}
void mwb_launch_reset(const MwbDev &d, int max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(reset_fn(d.task), dim3(d.N < max_blocks ? d.N : max_blocks), dim3(WAVE), mwb_reset_lds_bytes(d), s, d);
}
void mwb_launch_prep(const MwbDev &d, int mode, hipStream_t s) {
    const int blocks = mode == 1 ? 8 : (d.N + 255) / 256;
    hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, s, d, mode);
}
void mwb_launch_render(const MwbDev &d, int mode, hipStream_t s) {
    const dim3 g(d.N + d.split_envs), b(RENDER_THREADS);
    const size_t lds = mwb_render_lds_bytes(d);
    const dim3 gl(d.N < LIST_GRID ? d.N : LIST_GRID);
    if (d.tile_w > 0) {   // every frame as a few tiles, one workgroup each (see render_view_kernel): frames too large for one workgroup's LDS
        if (mode == 1) (void)launch_tiles<1>(d, d.tile_w, d.tile_h, s);
        else if (mode == 2) (void)launch_tiles<2>(d, d.tile_w, d.tile_h, s);
        else (void)launch_tiles<0>(d, d.tile_w, d.tile_h, s);
        return;
    }
    if (d.ent_task) {   // the general entity list: boxes, meshes, frames in up to MWB_MAX_ENTS slots
        if (mode == 1) render_kernel<RENDER_THREADS, 1, MWB_MAX_ENTS><<<gl, b, lds, s>>>(d);
        else if (mode == 2) render_kernel<RENDER_THREADS, 2, MWB_MAX_ENTS><<<g, b, lds, s>>>(d);
        else render_kernel<RENDER_THREADS, 0, MWB_MAX_ENTS><<<g, b, lds, s>>>(d);
        return;
    }
    if (d.poly) {   // YMaze: polygon rooms (one box)
        if (mode == 1) render_kernel<RENDER_THREADS, 1, 1, true><<<gl, b, lds, s>>>(d);
        else if (mode == 2) render_kernel<RENDER_THREADS, 2, 1, true><<<g, b, lds, s>>>(d);
        else render_kernel<RENDER_THREADS, 0, 1, true><<<g, b, lds, s>>>(d);
        return;
    }
    if (d.n_boxes == 2) {   // the two-box T-maze: its own instantiation, so that the one-box kernels stay as they are
        if (mode == 1) render_kernel<RENDER_THREADS, 1, 2><<<gl, b, lds, s>>>(d);
        else if (mode == 2) render_kernel<RENDER_THREADS, 2, 2><<<g, b, lds, s>>>(d);
        else render_kernel<RENDER_THREADS, 0, 2><<<g, b, lds, s>>>(d);
        return;
    }
    if (d.n_boxes == 6) {   // PutNext
        if (mode == 1) render_kernel<RENDER_THREADS, 1, 6><<<gl, b, lds, s>>>(d);
        else if (mode == 2) render_kernel<RENDER_THREADS, 2, 6><<<g, b, lds, s>>>(d);
        else render_kernel<RENDER_THREADS, 0, 6><<<g, b, lds, s>>>(d);
        return;
    }
    if (mode == 1) render_kernel<RENDER_THREADS, 1, 1><<<gl, b, lds, s>>>(d);
    else if (mode == 2) render_kernel<RENDER_THREADS, 2, 1><<<g, b, lds, s>>>(d);
    else render_kernel<RENDER_THREADS, 0, 1><<<g, b, lds, s>>>(d);
}
// ================================================================================== top view
// MiniWorldEnv.render_top_view (miniworld.py:1087-1158) for the whole batch: the floorplan from straight above (glOrtho over its
// extents + 1 m, widened to the frame's aspect), agent drawn.  Not on the training path - a display / debugging view, so
// simple: one workgroup per env, pixels strided over the threads, room tables read through L2.  Walls and box sides are
// edge-on and cover nothing; ceilings are back-face culled except those of rooms whose outline runs the other way round
// (they face up).  A sample sees the highest of: such a ceiling, the agent's triangle at agent.height (entity.py:494-514;
// lit with the normal the last box face left current, (0, -1, 0)), a box's top face, the floor of the first room holding the
// point, else the clear colour; equal heights go to what is drawn first (rooms, boxes in list order, agent).  Frozen float32
// choices as DESIGN.md 5.7: x = fmaf(wx, XS, X0), z = fmaf(-wy, ZS, Z1); inclusive containment tests; 8 coverage samples,
// one shade per (pixel, surface) at the pixel centre, LOD from the +1 pixel neighbours.
template <bool POLY>
__global__ void __launch_bounds__(256) top_view_kernel(MwbDev d, uint8_t *__restrict__ out, int W, int H) {
    __shared__ TexLds tex[MWB_MAX_TEX];
    const int e = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < (int)(sizeof(TexLds) / 4) * d.n_tex; i += 256) ((uint32_t *)tex)[i] = ((const uint32_t *)d.tex_desc)[i];
    __syncthreads();
    int n_rooms = d.n_rrooms[e];
    if (n_rooms < 0) n_rooms = 0;
    const float *rooms = d.rooms + (size_t)e * d.R_max * d.room_words;
    const float *fc = d.frame + (size_t)e * d.frame_words;
    double min_x = d.world_ext[e * 4 + 0] - 1, max_x = d.world_ext[e * 4 + 1] + 1, min_z = d.world_ext[e * 4 + 2] - 1, max_z = d.world_ext[e * 4 + 3] + 1;
    {
        const double width = max_x - min_x, height = max_z - min_z, aspect = width / height, fb_aspect = (double)W / (double)H;
        if (aspect > fb_aspect) { const double new_h = width / fb_aspect, h_diff = new_h - height; min_z -= h_diff / 2; max_z += h_diff / 2; }
        else if (aspect < fb_aspect) { const double new_w = height * fb_aspect, w_diff = new_w - width; min_x -= w_diff / 2; max_x += w_diff / 2; }
    }
    const float X0 = (float)min_x, XS = (float)((max_x - min_x) / (double)W), Z1 = (float)max_z, ZS = (float)((max_z - min_z) / (double)H);
    // the agent's triangle (float64 as the reference, then float32) and its height
    float tri[3][2];
    {
        const double adir = d.agent_dir[e], ar = d.agent_radius, ax = d.agent_x[e], az = d.agent_z[e];
        const double c = ref_cos(adir), s_ = ref_sin(adir);
        const double dvx = c * ar, dvz = -s_ * ar, rvx = s_ * ar, rvz = c * ar;
        tri[0][0] = (float)(ax + dvx); tri[0][1] = (float)(az + dvz);
        tri[1][0] = (float)(ax + 0.75 * (rvx - dvx)); tri[1][1] = (float)(az + 0.75 * (rvz - dvz));
        tri[2][0] = (float)(ax + 0.75 * (-rvx - dvx)); tri[2][1] = (float)(az + 0.75 * (-rvz - dvz));
    }
    const float agent_y = (float)(0.0 + 1.6);   // pos.y + Agent.height (entity.py:449)
    float lit_agent[3];
    {
        const float amb = (float)d.light_ambient[e * 3];
        const float v = (0.2f * 1.0f + amb * 1.0f) + 0.0f * (float)d.light_color[e * 3] * 1.0f;
        lit_agent[0] = v > 1.0f ? 1.0f : v; lit_agent[1] = 0.0f; lit_agent[2] = 0.0f;
    }
    enum { T_SKY = 0, T_FLOOR = 1, T_CEIL = 2, T_BOX = 4, T_AGENT = 5 };
    auto classify = [&](float x, float z, float &y_out) -> uint32_t {
        uint32_t key = T_SKY;
        float ybest = -INFINITY;
        for (int i = 0; i < n_rooms; i++) {
            bool in, culled = false;
            float height;
            if (!POLY) {
                const float4 rect = *(const float4 *)(rooms + i * MWB_ROOM_WORDS + RW_MINX);
                in = x >= rect.x && x <= rect.y && z >= rect.z && z <= rect.w;
                height = rooms[i * MWB_ROOM_WORDS + RW_HEIGHT];
            } else {
                const float *r = rooms + i * MWB_POLY_ROOM_WORDS;
                culled = ((__float_as_int(r[PW_FLAGS]) >> 8) & 1) != 0;
                const int ne = __float_as_int(r[PW_FLAGS]) & 255;
                height = r[PW_HEIGHT];
                in = true;
                for (int q = 0; q < ne; q++) {
                    const float *ed = r + PW_EDGE0 + PW_EDGE_WORDS * q;
                    const float side = fmaf(ed[5], z - ed[1], ed[4] * (x - ed[0]));
                    if (culled ? !(side <= 0.0f) : !(side >= 0.0f)) in = false;   // a reversed outline's normals point outwards
                }
            }
            if (!in) continue;
            if (!culled && 0.0f > ybest) { key = T_FLOOR | ((uint32_t)i << 3); ybest = 0.0f; }
            if (culled && !(height < 0.0f) && height > ybest) { key = T_CEIL | ((uint32_t)i << 3); ybest = height; }
        }
        for (int b = 0; b < d.n_boxes; b++) {
            const float *fb = fc + b * FC_BOX_STRIDE;
            const float rx = x - fb[FC_BOX_POS], rz = z - fb[FC_BOX_POS + 2];
            const float lx = rx * fb[FC_BOX_C] - rz * fb[FC_BOX_S], lz = rx * fb[FC_BOX_S] + rz * fb[FC_BOX_C];
            const float top = fb[FC_BOX_POS + 1] + fb[FC_BOX_SY];
            if (fabsf(lx) <= fb[FC_BOX_HX] && fabsf(lz) <= fb[FC_BOX_HZ] && top > ybest) { key = T_BOX | ((uint32_t)b << 3); ybest = top; }
        }
        {
            float sgn[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float *a = tri[q], *b = tri[(q + 1) % 3];
                sgn[q] = fmaf(b[0] - a[0], z - a[1], -((b[1] - a[1]) * (x - a[0])));
            }
            const bool in = (sgn[0] >= 0 && sgn[1] >= 0 && sgn[2] >= 0) || (sgn[0] <= 0 && sgn[1] <= 0 && sgn[2] <= 0);
            if (in && agent_y > ybest) { key = T_AGENT; ybest = agent_y; }
        }
        y_out = ybest;
        return key;
    };
    for (int p = tid; p < W * H; p += 256) {
        const int py = p / W, px = p - py * W;
        const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
        uint32_t keys[8];
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            const float wx = cx + c_sample_x[k], wy = cy + c_sample_y[k];
            float yy;
            keys[k] = classify(fmaf(wx, XS, X0), fmaf(-wy, ZS, Z1), yy);
        }
        const float xc = fmaf(cx, XS, X0), zc = fmaf(-cy, ZS, Z1);
        float acc[3] = {0, 0, 0};
        uint32_t done_mask = 0;
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            if (done_mask & (1u << k)) continue;
            int cnt = 0;
            for (int j = k; j < 8; j++)
                if (!(done_mask & (1u << j)) && keys[j] == keys[k]) { cnt++; done_mask |= 1u << j; }
            const uint32_t kind = keys[k] & 7u, idx = keys[k] >> 3;
            float col[3];
            if (kind == T_SKY) { col[0] = fc[FC_SKY]; col[1] = fc[FC_SKY + 1]; col[2] = fc[FC_SKY + 2]; }
            else if (kind == T_BOX) { const float *lb = fc + idx * FC_BOX_STRIDE + FC_LIT_BOX + 3 * 3; col[0] = lb[0]; col[1] = lb[1]; col[2] = lb[2]; }   // face 3 = +y
            else if (kind == T_AGENT) { col[0] = lit_agent[0]; col[1] = lit_agent[1]; col[2] = lit_agent[2]; }
            else {
                const uint32_t texw = (uint32_t)__float_as_int(rooms[idx * (POLY ? MWB_POLY_ROOM_WORDS : MWB_ROOM_WORDS) + (POLY ? PW_TEX : RW_TEX)]);
                const uint32_t tex_id = (kind == T_FLOOR ? (texw >> 8) : (texw >> 16)) & 255u;
                const TexLds &T = tex[tex_id];
                const float *lit = fc + (kind == T_FLOOR ? FC_LIT_FLOOR : FC_LIT_CEIL);
                float texel[3];
                sample_texture(d.texels, T, xc * T.sc_s, zc * T.sc_t, (xc + XS) * T.sc_s, zc * T.sc_t, xc * T.sc_s, (zc - ZS) * T.sc_t, true, texel);
                for (int q = 0; q < 3; q++) col[q] = lit[q] * (texel[q] * (1.0f / 255.0f));
            }
            for (int q = 0; q < 3; q++) acc[q] += (float)cnt * col[q];
        }
        uint8_t *o = out + ((size_t)e * W * H + p) * 3;
        for (int q = 0; q < 3; q++) {
            float v = acc[q] * 0.125f;
            v = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
            o[q] = (uint8_t)(unsigned)(v * 255.0f + 0.5f);
        }
    }
}
void mwb_launch_top_view(const MwbDev &d, uint8_t *out, int W, int H, hipStream_t s) {
    if (d.poly) top_view_kernel<true><<<dim3(d.N), dim3(256), 0, s>>>(d, out, W, H);
    else top_view_kernel<false><<<dim3(d.N), dim3(256), 0, s>>>(d, out, W, H);
}

// =========================================================================== get_visible_ents
// MiniWorldEnv.get_visible_ents (miniworld.py:1222-1315) for the whole batch: rooms into the observation frame (8 samples, the
// camera of render_obs), then per entity but the agent, in list order, an axis-aligned 0.2 m cube at its position inside a
// GL_ANY_SAMPLES_PASSED query; GL_LESS with depth writes on, so a cube is visible iff at some sample its front face is nearer
// than the room surface and than every cube drawn before it (ray parameters of the same ray compared; depth-buffer
// quantisation and the near plane are not modelled).  One workgroup per env, pixels strided over the threads; a sample that
// meets no cube - nearly all of them - costs the slab tests only.  Not on the training path; the reference never calls it.
template <bool POLY>
__global__ void __launch_bounds__(256) visible_kernel(MwbDev d, uint32_t *__restrict__ mask_out) {
    __shared__ int cam_room_s;
    __shared__ uint32_t mask_s;
    const int e = blockIdx.x, tid = threadIdx.x, W = d.W, H = d.H;
    int n_rooms = d.n_rrooms[e];
    if (n_rooms < 0) n_rooms = 0;
    const float *rooms = d.rooms + (size_t)e * d.R_max * d.room_words;
    const float *fc = d.frame + (size_t)e * d.frame_words;
    if (tid == 0) { cam_room_s = 0x7fffffff; mask_s = 0; }
    __syncthreads();
    Cam cam;
    for (int k = 0; k < 3; k++) { cam.eye[k] = fc[FC_EYE + k]; cam.F[k] = fc[FC_F + k]; cam.S[k] = fc[FC_S + k]; cam.U[k] = fc[FC_U + k]; }
    cam.TW = fc[FC_TW]; cam.TH = fc[FC_TH];
    cam.Wf = (float)W; cam.Hf = (float)H; cam.invW = 1.0f / (float)W; cam.invH = 1.0f / (float)H;
    for (int i = tid; i < n_rooms; i += 256) {
        bool in;
        if (POLY) {
            const float *r = rooms + i * MWB_POLY_ROOM_WORDS;
            in = !((__float_as_int(r[PW_FLAGS]) >> 8) & 1);
            for (int k = 0; k < 4; k++) {
                const float *ed = r + PW_EDGE0 + PW_EDGE_WORDS * k;
                if (!(fmaf(ed[5], cam.eye[2] - ed[1], ed[4] * (cam.eye[0] - ed[0])) >= 0.0f)) in = false;
            }
        } else {
            const float *r = rooms + i * MWB_ROOM_WORDS;
            in = cam.eye[0] >= r[RW_MINX] && cam.eye[0] <= r[RW_MAXX] && cam.eye[2] >= r[RW_MINZ] && cam.eye[2] <= r[RW_MAXZ];
        }
        if (in) atomicMin(&cam_room_s, i);
    }
    __syncthreads();
    const int cam_room = cam_room_s == 0x7fffffff ? -1 : cam_room_s;
    // the cubes in the order the reference draws them: self.entities without the agent (entity tasks: the env's order list)
    float lo[MWB_MAX_ENTS][3], hi[MWB_MAX_ENTS][3];
    int slot[MWB_MAX_ENTS];
    int n_cubes = 0;
    if (d.ent_task) {
        const int n_ord = d.n_order[e];
        for (int q = 0; q < n_ord && n_cubes < MWB_MAX_ENTS; q++) {
            const int sl = d.ent_order[(size_t)e * MWB_ORDER_STRIDE + q];
            if (sl != MWB_ENT_AGENT) slot[n_cubes++] = sl;
        }
    } else {
        for (int b = 0; b < d.n_boxes; b++) slot[n_cubes++] = b;
    }
    for (int b = 0; b < n_cubes; b++) {   // glVertex3f arguments: float32 of the float64 sums
        const size_t be = (size_t)slot[b] * d.N + e;
        const double px = d.box_x[be], py = d.box_y[be], pz = d.box_z[be];
        lo[b][0] = (float)(px - 0.1); hi[b][0] = (float)(px + 0.1);
        lo[b][1] = (float)py;         hi[b][1] = (float)(py + 0.2);
        lo[b][2] = (float)(pz - 0.1); hi[b][2] = (float)(pz + 0.1);
    }
    uint32_t mask = 0;
    for (int p = tid; p < W * H; p += 256) {
        const int py = p / W, px = p - py * W;
        const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            float dv[3], tb[MWB_MAX_ENTS];
            make_ray(cam, cx + c_sample_x[k], cy + c_sample_y[k], dv);
            bool any = false;
            for (int b = 0; b < n_cubes; b++) {
                float tn = -INFINITY, tf = INFINITY;
                bool miss = false;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    if (dv[a] == 0) { miss = miss || cam.eye[a] < lo[b][a] || cam.eye[a] > hi[b][a]; continue; }
                    const float inv = 1.0f / dv[a];
                    const float t1 = (lo[b][a] - cam.eye[a]) * inv, t2 = (hi[b][a] - cam.eye[a]) * inv;
                    const float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
                    if (tmin > tn) tn = tmin;
                    if (tmax < tf) tf = tmax;
                }
                const bool hit = !miss && tn <= tf && tn > 0;   // eye inside the cube: only back faces, culled
                tb[b] = hit ? tn : INFINITY;
                any = any || hit;
            }
            if (!any) continue;
            float th;
            uint32_t path;
            const uint32_t key = POLY ? trace_rooms_poly<false>(rooms, n_rooms, cam_room, cam.eye, dv, th, path)
                                      : trace_rooms<false>(rooms, n_rooms, cam_room, cam.eye, dv, th, path);
            float depth = (key & 7u) != KIND_SKY ? th : INFINITY;   // the depth buffer at this sample so far
            for (int b = 0; b < n_cubes; b++)
                if (tb[b] < depth) { mask |= 1u << slot[b]; depth = tb[b]; }
        }
    }
    if (mask) atomicOr(&mask_s, mask);
    __syncthreads();
    if (tid == 0) mask_out[e] = mask_s;
}
void mwb_launch_visible(const MwbDev &d, uint32_t *mask_out, hipStream_t s) {
    if (d.poly) visible_kernel<true><<<dim3(d.N), dim3(256), 0, s>>>(d, mask_out);
    else visible_kernel<false><<<dim3(d.N), dim3(256), 0, s>>>(d, mask_out);
}

void mwb_launch_intersect(const MwbDev &d, int env, int ent, double x, double z, double radius, int *result_dev, hipStream_t s) {
    hipLaunchKernelGGL(intersect_kernel, dim3(1), dim3(64), 0, s, d, env, ent, x, z, radius, result_dev);
}
