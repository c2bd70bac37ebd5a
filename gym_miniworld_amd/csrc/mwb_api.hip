// mwb_api.hip - host side of the C ABI declared in include/miniworld_batch.h.
//
// Owns the HBM-resident state of one shard of environments, hashes seeds to MT19937 states,
// builds the texture mip pyramids, and enqueues the step / reset / prep / render kernels on the
// caller's stream.  No torch types, no oracle code; fails loudly (negative status + message)
// rather than falling back to any CPU path.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mwb_internal.h"

static thread_local std::string g_err;
static int set_err(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return set_err(MWB_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

// Every entry point runs on the handle's device and puts the caller's current device back on the way out
// (torch.cuda.current_device() reads the runtime's current device: a handle on cuda:1 must not move it).
struct DevGuard {
    int prev = -1;
    hipError_t err;
    explicit DevGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev); else if (err == hipSuccess) prev = -1;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define USE_DEVICE(dev)                                                                             \
    DevGuard _dev_guard(dev);                                                                        \
    if (_dev_guard.err != hipSuccess)                                                                \
        return set_err(MWB_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(_dev_guard.err))

struct mwb_handle {
    mwb_config cfg;
    MwbDev dev;
    std::vector<void *> allocs;
    std::vector<std::vector<uint32_t>> tex_levels[MWB_MAX_TEX];   // host mip chains (RGBA8 packed)
    int tex_w[MWB_MAX_TEX], tex_h[MWB_MAX_TEX];
    uint32_t *texels_dev;
    MwbTexDesc *tex_desc_dev;
    bool seeded, textures_dirty, have_textures;
    bool have_obs;   // some pass has rendered an observation since creation
    int *scratch_int_dev;
    // timing
    // timing: five events per pipeline pass, drawn from a pool and only read back in
    // mwb_timing_read, so that enabling it adds no host synchronisation to the timed region
    bool timing, timing_now;   // enabled at all / recording the current pass
    int timing_period, timing_tick;   // record every timing_period-th pass (the events themselves cost ~20 us per pass)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used;
    hipEvent_t *ev;   // the current pass' events: 0-4 on the caller's stream, 5-6 around reset_kernel
    // mwb_step overlaps world generation of the finished envs with the bulk render on a side stream
    bool overlap_reset;
    void *pack;          // done | reward | feature | goal_pos | ep_steps | reward64 in one allocation
    size_t pack_bytes;
    void *stack;
    int stack_n, stack_dtype;
    int stack_fused;               // the render kernels write the frames into the window themselves (MWB_STACK_FUSED)
    int stack_planes, stack_pos;   // sliding-window stack: planes per env (0 = the classic shifting stack), the window's first plane
    size_t stack_bytes;
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    // entity tasks: mesh geometries (host copies until the first pass uploads them in one allocation)
    struct HostMesh { bool set = false; int n_tris = 0, n_nodes = 0, n_orders = 1, tex_id = -1; float min_c[3], max_c[3]; std::vector<float> nodes, tris, tris2, shade; };
    HostMesh meshes[MWB_NUM_MESHES];
    bool meshes_dirty;
    float4 *mesh_data_dev;
    MwbMeshDesc *mesh_desc_dev;
    float *view_frame;   // frame constants of mwb_render_view's size (lazily allocated)
};

extern "C" const char *mwb_last_error(void) { return g_err.c_str(); }
extern "C" int mwb_abi_version(void) { return MWB_ABI_VERSION; }

// ------------------------------------------------------------------------------------ SHA-512
// Needed for the seed -> MT19937 key mapping of gym<=0.21 (gym/utils/seeding.py hash_seed, an
// un-vendored dependency of reference random.py:10): key = little-endian 32-bit limbs of the first
// 8 bytes of sha512(str(seed)).  "Parity unpinned": gym is not available to check against.
static const uint64_t K512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void sha512_short(const uint8_t *msg, size_t len, uint8_t out[64]) {   // len < 112: one block
    uint8_t blk[128];
    memset(blk, 0, sizeof(blk));
    memcpy(blk, msg, len);
    blk[len] = 0x80;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) blk[127 - i] = (uint8_t)(bits >> (8 * i));
    uint64_t Hs[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                      0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint64_t w[80];
    for (int i = 0; i < 16; i++) {
        w[i] = 0;
        for (int k = 0; k < 8; k++) w[i] = (w[i] << 8) | blk[i * 8 + k];
    }
    for (int i = 16; i < 80; i++) {
        uint64_t s0 = rotr64(w[i - 15], 1) ^ rotr64(w[i - 15], 8) ^ (w[i - 15] >> 7);
        uint64_t s1 = rotr64(w[i - 2], 19) ^ rotr64(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint64_t a = Hs[0], b = Hs[1], c = Hs[2], d = Hs[3], e = Hs[4], f = Hs[5], g = Hs[6], h = Hs[7];
    for (int i = 0; i < 80; i++) {
        uint64_t S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
        uint64_t ch = (e & f) ^ (~e & g);
        uint64_t t1 = h + S1 + ch + K512[i] + w[i];
        uint64_t S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
        uint64_t maj = (a & b) ^ (a & c) ^ (b & c);
        uint64_t t2 = S0 + maj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    Hs[0] += a; Hs[1] += b; Hs[2] += c; Hs[3] += d; Hs[4] += e; Hs[5] += f; Hs[6] += g; Hs[7] += h;
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 8; k++) out[i * 8 + k] = (uint8_t)(Hs[i] >> (56 - 8 * k));
}

// seed (already reduced mod 2^64 by the uint64 type) -> MT19937 init_by_array key
static int seed_to_key(uint64_t seed, uint32_t key[2]) {
    char txt[32];
    int n = snprintf(txt, sizeof(txt), "%llu", (unsigned long long)seed);
    uint8_t dig[64];
    sha512_short((const uint8_t *)txt, (size_t)n, dig);
    uint32_t lo = dig[0] | (dig[1] << 8) | (dig[2] << 16) | ((uint32_t)dig[3] << 24);
    uint32_t hi = dig[4] | (dig[5] << 8) | (dig[6] << 16) | ((uint32_t)dig[7] << 24);
    key[0] = lo; key[1] = hi;
    return hi != 0 ? 2 : 1;   // _int_list_from_bigint drops the high limb when it is zero ([0] for 0)
}

extern "C" int mwb_seed_key(uint64_t seed, uint32_t key[2]) { return seed_to_key(seed, key); }

// numpy RandomState.seed(list) == MT19937 init_by_array
static void mt_init_by_array(uint32_t *mt, const uint32_t *init_key, int key_length) {
    mt[0] = 19650218u;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int i = 1, j = 0;
    for (int k = (624 > key_length ? 624 : key_length); k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + init_key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
    mt[624] = 624;   // position: the first draw regenerates the state
}

// ------------------------------------------------------------------------------------ helpers
template <typename T>
static int dev_alloc(mwb_handle *h, T **p, size_t n) {
    void *q = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return set_err(MWB_ENOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return set_err(MWB_EHIP, std::string("hipMemset failed: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return MWB_OK;
}

static void default_params(double p[MWB_NPARAM][9]) {   // params.py:110-123
    static const double T[MWB_NPARAM][9] = {
        {0.25, 0.82, 1, 0.1, 0.1, 0.1, 1.0, 1.0, 1.0},       {0, 2.5, 0, -40, 2.5, -40, 40, 5, 40},
        {0.7, 0.7, 0.7, 0.45, 0.45, 0.45, 0.8, 0.8, 0.8},    {0.45, 0.45, 0.45, 0.35, 0.35, 0.35, 0.55, 0.55, 0.55},
        {0, 0, 0, -0.2, -0.2, -0.2, 0.2, 0.2, 0.2},          {0.15, 0, 0, 0.12, 0, 0, 0.17, 0, 0},
        {0, 0, 0, -0.05, 0, 0, 0.05, 0, 0},                  {15, 0, 0, 10, 0, 0, 20, 0, 0},
        {0.4, 0, 0, 0.38, 0, 0, 0.42, 0, 0},                 {0, 0, 0, -5, 0, 0, 5, 0, 0},
        {60, 0, 0, 55, 0, 0, 65, 0, 0},                      {1.5, 0, 0, 1.45, 0, 0, 1.55, 0, 0},
        {0, 0, 0, -0.05, 0, 0, 0.10, 0, 0}};
    memcpy(p, T, sizeof(T));
}

extern "C" int mwb_create(const mwb_config *cfg, mwb_handle **out) {
    if (!cfg || !out) return set_err(MWB_EINVAL, "mwb_create: null argument");
    if (cfg->abi_version != MWB_ABI_VERSION) return set_err(MWB_EINVAL, "mwb_create: abi_version mismatch");
    if (cfg->num_envs <= 0) return set_err(MWB_EINVAL, "mwb_create: num_envs must be > 0");
    if (cfg->task < 0 || cfg->task >= MWB_NUM_TASKS) return set_err(MWB_EINVAL, "mwb_create: unknown task");
    if (cfg->obs_width <= 0 || cfg->obs_height <= 0 || cfg->obs_width > 1024 || cfg->obs_height > 1024)
        return set_err(MWB_EINVAL, "mwb_create: bad observation size");
    if (cfg->layout != MWB_LAYOUT_HWC && cfg->layout != MWB_LAYOUT_CWH) return set_err(MWB_EINVAL, "mwb_create: bad layout");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_err(MWB_EHIP, "mwb_create: no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return set_err(MWB_EINVAL, "mwb_create: bad device ordinal");
    USE_DEVICE(cfg->device);

    mwb_handle *h = new mwb_handle();
    h->cfg = *cfg;
    MwbDev &d = h->dev;
    memset(&d, 0, sizeof(d));
    d.N = cfg->num_envs; d.task = cfg->task; d.W = cfg->obs_width; d.H = cfg->obs_height;
    d.want_depth = cfg->want_depth ? 1 : 0; d.layout = cfg->layout; d.domain_rand = cfg->domain_rand ? 1 : 0;
    d.auto_reset = cfg->no_auto_reset ? 0 : 1;
    { const char *dbg = getenv("MWB_DEBUG"); d.debug_flags = dbg ? atoi(dbg) : 0; }
    { const char *ex = getenv("MWB_EXP"); d.exp_flags = ex ? atoi(ex) : 0; }
    d.tile_w = 0; d.tile_h = 0;
    if (cfg->task >= MWB_TASK_PICKUPOBJS) {   // MWB_TILE=WxH (0x0: whole frames): the tile a workgroup renders in the entity tasks
        int tw_ = 0, th_ = 0;   // default: whole frames (tiles cost 1.3 - 2.7x more in total: measured, scripts/ab_mesh_phases.py)
        const char *tl = getenv("MWB_TILE");
        if (tl && sscanf(tl, "%dx%d", &tw_, &th_) != 2) { tw_ = 0; th_ = 0; }
        if (tw_ > 0 && th_ > 0) { d.tile_w = tw_ < cfg->obs_width ? tw_ : cfg->obs_width; d.tile_h = th_ < cfg->obs_height ? th_ : cfg->obs_height; }
    }
    static const double dflt[MWB_NUM_TASKS][4] = {{12, 0, 0, 0}, {10, 0, 0, 0}, {0, 0, 0, 0}, {8, 8, 3, 0}, {0, 0, 0, 0}, {0, 0, 0, 100}, {0, 0, 0, 0}, {0, 0, 0, 0}, {12, 0, 0, 0}, {0, 0, 0, 0},
                                                 {12, 5, 0, 0}, {10, 0, 0, 0}, {16, 0, 0, 0}, {0, 0, 0, 0}, {10, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool sim2real = cfg->task == MWB_TASK_SIM2REAL_GOTO || cfg->task == MWB_TASK_SIM2REAL_PUSH;
    const int task = cfg->task;
    d.ent_task = task >= MWB_TASK_PICKUPOBJS ? 1 : 0;
    d.n_boxes = (cfg->task == MWB_TASK_TMAZE_TWOBOX || cfg->task == MWB_TASK_SIM2REAL_PUSH) ? 2 : cfg->task == MWB_TASK_PUTNEXT ? 6 : 1;
    if (task == MWB_TASK_PICKUPOBJS) {
        const double no = cfg->task_args[1] != 0 ? cfg->task_args[1] : 5;
        if (!(no >= 1 && no <= MWB_MAX_ENTS) || no != (double)(int)no) { delete h; return set_err(MWB_EINVAL, "PickupObjs: num_objs must be 1 .. 20"); }
        d.n_boxes = (int)no;
    } else if (task == MWB_TASK_ROOMOBJS) d.n_boxes = 3;
    else if (task == MWB_TASK_COLLECTHEALTH) d.n_boxes = 18;
    else if (task == MWB_TASK_THREEROOMS) d.n_boxes = 6;
    else if (task == MWB_TASK_SIGN || task == MWB_TASK_SIDEWALK) d.n_boxes = 7;
    else if (task == MWB_TASK_WALLGAP) d.n_boxes = 2;
    d.frame_words = MWB_FRAME_WORDS_FOR(d.ent_task ? MWB_MAX_ENTS : d.n_boxes);   // the entity render kernels are compiled for every slot
    d.n_tex = sim2real ? 17 : d.ent_task ? (task == MWB_TASK_SIGN ? MWB_NUM_TEXTURES : MWB_TEX_CHAR0) : 7;
    {   // the last ~0.6 round of resident workgroups (5 per CU x 256 CUs = 1280) of a bulk render launch drains in
        // half-frame units: measured best between 640 and 960 envs (+4 % at 8192 Maze envs); MWB_SPLIT overrides
        const char *sp = getenv("MWB_SPLIT");
        int split = sp ? atoi(sp) : 768;
        d.split_envs = split < 0 ? 0 : (split > cfg->num_envs ? cfg->num_envs : split);
    }
    d.no_ceiling = (sim2real || task == MWB_TASK_PICKUPOBJS || task == MWB_TASK_ROOMOBJS || task == MWB_TASK_SIDEWALK || task == MWB_TASK_WALLGAP) ? 1 : 0;
    d.poly = cfg->task == MWB_TASK_YMAZE ? 1 : 0;
    d.room_words = d.poly ? MWB_POLY_ROOM_WORDS : MWB_ROOM_WORDS;
    d.agent_radius = sim2real ? 0.11 : task == MWB_TASK_ROOMOBJS ? 1.5 : 0.4;   // simtorealgoto.py:50 / roomobjs.py:36 / entity.py:451
    for (int i = 0; i < 4; i++) d.task_args[i] = cfg->task_args[i] != 0 ? cfg->task_args[i] : dflt[cfg->task][i];
    int mes = cfg->max_episode_steps;
    if (mes <= 0) {   // hallway.py:18, oneroom.py:14, fourrooms.py:15, maze.py:27
        if (d.task == MWB_TASK_HALLWAY) mes = 250;
        else if (d.task == MWB_TASK_ONEROOM) mes = 180;
        else if (d.task == MWB_TASK_FOURROOMS) mes = 250;
        else if (d.task == MWB_TASK_TMAZE || d.task == MWB_TASK_TMAZE_TWOBOX) mes = 280;   // tmaze.py:20,142
        else if (d.task == MWB_TASK_SIM2REAL_GOTO) mes = 100;   // simtorealgoto.py:30
        else if (d.task == MWB_TASK_SIM2REAL_PUSH) mes = 150;   // simtorealpush.py:29
        else if (d.task == MWB_TASK_PUTNEXT) mes = 250;   // putnext.py:16
        else if (d.task == MWB_TASK_YMAZE) mes = 280;     // ymaze.py:21
        else if (d.task == MWB_TASK_PICKUPOBJS) mes = 400;        // pickupobjs.py:19
        else if (d.task == MWB_TASK_ROOMOBJS) mes = 2147483647;   // roomobjs.py:19: math.inf
        else if (d.task == MWB_TASK_COLLECTHEALTH) mes = 1000;    // collecthealth.py:24
        else if (d.task == MWB_TASK_THREEROOMS) mes = 400;        // threerooms.py:14
        else if (d.task == MWB_TASK_SIGN) mes = 20;               // sign.py:41
        else if (d.task == MWB_TASK_SIDEWALK) mes = 150;          // sidewalk.py:15
        else if (d.task == MWB_TASK_WALLGAP) mes = 300;           // wallgap.py:14
        else mes = (int)d.task_args[0] * (int)d.task_args[1] * 24;
    }
    d.max_episode_steps = mes;
    double P[MWB_NPARAM][9];
    if (cfg->use_default_params) default_params(P); else memcpy(P, cfg->params, sizeof(P));
    for (int i = 0; i < MWB_NPARAM; i++)
        for (int k = 0; k < 3; k++) { d.params[i].def[k] = P[i][k]; d.params[i].lo[k] = P[i][3 + k]; d.params[i].hi[k] = P[i][6 + k]; }
    if (d.task == MWB_TASK_HALLWAY) { if (!(d.task_args[0] >= 2)) { delete h; return set_err(MWB_EINVAL, "Hallway: length >= 2"); } d.R_max = 1; d.S_max = 4; }
    else if (d.task == MWB_TASK_ONEROOM) { if (!(d.task_args[0] >= 2)) { delete h; return set_err(MWB_EINVAL, "OneRoom: size >= 2"); } d.R_max = 1; d.S_max = 4; }
    else if (d.task == MWB_TASK_FOURROOMS) { d.R_max = 8; d.S_max = 32; }
    else if (d.task == MWB_TASK_TMAZE || d.task == MWB_TASK_TMAZE_TWOBOX) {
        if (d.task_args[3] < 0 || d.task_args[3] > 9e15) { delete h; return set_err(MWB_EINVAL, "TMaze: bad sub_task_length"); }
        d.R_max = 2; d.S_max = 8;
    } else if (sim2real) { d.R_max = 1; d.S_max = 4; }
    else if (d.task == MWB_TASK_YMAZE) { d.R_max = 6; d.S_max = 24; }   // corridor, hub, two arms, two connectors (the third pair of portals meets directly)
    else if (d.task == MWB_TASK_PUTNEXT) { if (!(d.task_args[0] >= 2)) { delete h; return set_err(MWB_EINVAL, "PutNext: size >= 2"); } d.R_max = 1; d.S_max = 4; }
    else if (d.task == MWB_TASK_PICKUPOBJS || d.task == MWB_TASK_ROOMOBJS || d.task == MWB_TASK_COLLECTHEALTH) {
        if (!(d.task_args[0] >= 2)) { delete h; return set_err(MWB_EINVAL, "size >= 2"); }
        d.R_max = 1; d.S_max = 4;
    } else if (d.ent_task) {
        if (d.task == MWB_TASK_SIGN && (d.task_args[0] != 10 || d.task_args[1] < 0 || d.task_args[1] > 2 || d.task_args[2] < 0 || d.task_args[2] > 1 ||
                                        d.task_args[1] != (double)(int)d.task_args[1] || d.task_args[2] != (double)(int)d.task_args[2])) {
            delete h;   // the objects sit at fixed coordinates (sign.py:91-102): only the default room fits them
            return set_err(MWB_EINVAL, "Sign: size must be 10, color_index 0..2, goal 0..1");
        }
        d.R_max = 8; d.S_max = 24;
    }
    else {
        int rows = (int)d.task_args[0], cols = (int)d.task_args[1];
        if (rows < 1 || cols < 1 || rows * cols > 4096) { delete h; return set_err(MWB_EINVAL, "Maze: bad num_rows / num_cols"); }
        d.R_max = 2 * rows * cols - 1; d.S_max = 4 * rows * cols;
        if (d.S_max < 8) d.S_max = 8;
    }
    size_t N = (size_t)d.N;
    int rc = MWB_OK;
#define A(ptr, n) if (rc == MWB_OK) rc = dev_alloc(h, &(ptr), (n))
    A(d.agent_x, N); A(d.agent_z, N); A(d.agent_dir, N); A(d.box_x, N * d.n_boxes); A(d.box_z, N * d.n_boxes); A(d.box_y, N * d.n_boxes); A(d.box_dir, N * d.n_boxes); A(d.carrying, N);
    A(d.box_color, N * d.n_boxes * 3); A(d.box_size, N * d.n_boxes); A(d.goal_dist, N); A(d.episode_count, N); A(d.task_step_count, N); A(d.goal_idx, N); A(d.cam, N * 4); A(d.sky_color, N * 3); A(d.light_pos, N * 3); A(d.light_color, N * 3); A(d.light_ambient, N * 3);
    A(d.step_count, N); A(d.n_rooms, N); A(d.n_rrooms, N); A(d.seg_stage, (size_t)MWB_RESET_MAX_BLOCKS * d.S_max * 4); A(d.n_segs, N); A(d.reset_set, N); A(d.reset_list, N); A(d.reset_count, (size_t)1);
    if (d.ent_task) {
        A(d.ent_meta, N * d.n_boxes); A(d.ent_radius, N * d.n_boxes); A(d.ent_height, N * d.n_boxes); A(d.ent_scale, N * d.n_boxes);
        A(d.ent_order, N * MWB_ORDER_STRIDE); A(d.n_order, N); A(d.task_f, N); A(d.task_i, N); A(d.text_tex, N * 8);
        A(d.ovr_slot, N); A(d.ovr_pose, N * 4);
        A(h->mesh_desc_dev, (size_t)MWB_NUM_MESHES);
    }
    A(d.rng, N * MWB_MT_WORDS); A(d.rooms, N * d.R_max * d.room_words); A(d.segs, N * d.S_max * 4); A(d.frame, N * d.frame_words); A(d.world_ext, N * 4);
    A(d.obs, N * d.W * d.H * 3);
    if (d.want_depth) { A(d.depth, N * d.W * d.H); }
    {   // the small per-step outputs share one allocation (one D2H copy for a host-side consumer); 16-byte aligned parts
        auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
        // ordered by how often a host-side consumer needs them, so that it can copy a PREFIX: done | reward (what every step of a
        // VecEnv returns) | feature | goal_pos (the T-/Y-maze infos) | ep_steps | reward64
        const size_t o_done = 0, o_rew = o_done + up(N), o_feat = o_rew + up(N * 4), o_goal = o_feat + up(N * 8),
                     o_eps = o_goal + up(N * 24), o_r64 = o_eps + up(N * 4), total = o_r64 + up(N * 8);
        uint8_t *pack = nullptr;
        A(pack, total);
        if (rc == MWB_OK) {
            h->pack = pack; h->pack_bytes = total;
            d.reward64 = (double *)(pack + o_r64); d.goal_pos = (double *)(pack + o_goal); d.reward = (float *)(pack + o_rew);
            d.feature = (float *)(pack + o_feat); d.ep_steps = (int32_t *)(pack + o_eps); d.done = pack + o_done;
        }
    }
    A(h->tex_desc_dev, (size_t)MWB_MAX_TEX);
    A(h->scratch_int_dev, (size_t)4);
    A(d.error_flag, (size_t)1);
#undef A
    if (rc == MWB_OK && (d.debug_flags & 16)) rc = dev_alloc(h, &d.wg_ts, 2 * (N + (size_t)d.split_envs));
    if (rc == MWB_OK && (d.exp_flags & 4)) rc = dev_alloc(h, &d.dbg_counters, (size_t)8);
    if (rc == MWB_OK) rc = dev_alloc(h, &d.cost, 2 * N);
    if (rc == MWB_OK) rc = dev_alloc(h, &d.bucket, N);
    if (rc == MWB_OK) rc = dev_alloc(h, &d.order_bufs[0], N);
    if (rc == MWB_OK) rc = dev_alloc(h, &d.order_bufs[1], N);
    if (rc == MWB_OK) rc = dev_alloc(h, &d.order_state, (size_t)2);
    if (rc == MWB_OK) {   // no cost known yet: identity order
        std::vector<int32_t> ident(N);
        for (size_t i = 0; i < N; i++) ident[i] = (int32_t)i;
        for (int k = 0; k < 2 && rc == MWB_OK; k++)
            if (hipMemcpy(d.order_bufs[k], ident.data(), N * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) rc = set_err(MWB_EHIP, "mwb_create: hipMemcpy failed");
    }
    if (rc != MWB_OK) { mwb_destroy(h); return rc; }
    // MiniWorldEnv.__init__ ends with self.reset() (miniworld.py:523): the episode counters of TMazeDynamic /
    // TMazeTwoBoxDynamic (tmaze.py:80,130) have seen one reset when the caller gets the env
    if ((d.task == MWB_TASK_TMAZE && d.task_args[3] > 0) || (d.task == MWB_TASK_TMAZE_TWOBOX && d.task_args[0] == 0)) {
        std::vector<int64_t> ones(N, 1);
        if (hipMemcpy(d.episode_count, ones.data(), N * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) {
            mwb_destroy(h);
            return set_err(MWB_EHIP, "mwb_create: hipMemcpy failed");
        }
    }
    if (hipMemset(d.carrying, 0xFF, N * sizeof(int32_t)) != hipSuccess) { mwb_destroy(h); return set_err(MWB_EHIP, "mwb_create: hipMemset failed"); }   // -1: nothing carried
    d.tex_desc = h->tex_desc_dev;
    h->texels_dev = nullptr; d.texels = nullptr;
    h->seeded = false; h->textures_dirty = false; h->have_textures = false; h->have_obs = false;
    h->timing = false; h->timing_now = false; h->timing_period = 1; h->timing_tick = 0; h->ev_used = 0; h->ev = nullptr;
    { const char *no = getenv("MWB_NO_OVERLAP"); h->overlap_reset = !(no && atoi(no)); }
    h->side = nullptr; h->ev_fork = nullptr; h->ev_join = nullptr;
    h->view_frame = nullptr;
    h->meshes_dirty = false; h->mesh_data_dev = nullptr; d.mesh_desc = h->mesh_desc_dev; d.mesh_data = nullptr;
    h->stack = nullptr; h->stack_n = 0; h->stack_dtype = 0; h->stack_bytes = 0; h->stack_planes = 0; h->stack_pos = 0; h->stack_fused = 0;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        mwb_destroy(h);
        return set_err(MWB_EHIP, "mwb_create: could not create the side stream / events");
    }
    for (int i = 0; i < MWB_MAX_TEX; i++) { h->tex_w[i] = 0; h->tex_h[i] = 0; }
    if (d.tile_w == 0) {   // an observation that does not fit one workgroup's LDS (or its 16-bit pixel queue) is rendered in tiles, like mwb_render_view's frames
        int wshift = 0;
        while ((1 << wshift) < d.W) wshift++;
        if (mwb_render_lds_bytes(d) > 160 * 1024 || ((size_t)d.H << wshift) > 65536) {
            int tw = 0, th = 0;
            mwb_view_tile(&tw, &th);
            d.tile_w = tw < d.W ? tw : d.W; d.tile_h = th < d.H ? th : d.H;
        }
    }
    if (int prc = mwb_prepare_kernels(d)) {
        mwb_destroy(h);
        return set_err(prc == -2 ? MWB_EHIP : MWB_EINVAL,
                       prc == -1 ? "mwb_create: world + observation too large for the 160 KB LDS staging buffers (room table + W*H*3 frame)"
                       : prc == -3 ? "mwb_create: observation too large for the 16-bit pixel queue (H << ceil(log2 W) must not exceed 65536)"
                                   : "mwb_create: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
    *out = h;
    return MWB_OK;
}

extern "C" int mwb_destroy(mwb_handle *h) {
    if (!h) return MWB_OK;
    DevGuard _dev_guard(h->cfg.device);
    hipDeviceSynchronize();
    for (void *p : h->allocs) hipFree(p);
    if (h->texels_dev) hipFree(h->texels_dev);
    if (h->mesh_data_dev) hipFree(h->mesh_data_dev);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    delete h;
    return MWB_OK;
}

// ------------------------------------------------------------------------------------ textures
// Mip chain: level k+1 has dims max(1, n/2); each texel is the equal-weight mean (round half up) of
// the source texels it covers - 2x2 for even sizes (DESIGN.md render spec; restates
// glGenerateMipmap of opengl.py:98-99, whose filter GL leaves to the driver).
static void build_mips(const uint8_t *rgb, int w, int h, std::vector<std::vector<uint32_t>> &levels) {
    levels.clear();
    std::vector<uint32_t> cur((size_t)w * h);
    for (int y = 0; y < h; y++)   // flip: texture row 0 = bottom image row (pyglet upload order, opengl.py:85-96)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = rgb + ((size_t)(h - 1 - y) * w + x) * 3;
            cur[(size_t)y * w + x] = p[0] | (p[1] << 8) | (p[2] << 16) | 0xFF000000u;
        }
    levels.push_back(cur);
    int sw = w, sh = h;
    while (sw > 1 || sh > 1) {
        int dw = sw > 1 ? sw / 2 : 1, dh = sh > 1 ? sh / 2 : 1;
        std::vector<uint32_t> nxt((size_t)dw * dh);
        const std::vector<uint32_t> &src = levels.back();
        for (int j = 0; j < dh; j++) {
            int j0 = (int)(((long long)j * sh) / dh), j1 = (int)((((long long)(j + 1) * sh) + dh - 1) / dh);
            for (int i = 0; i < dw; i++) {
                int i0 = (int)(((long long)i * sw) / dw), i1 = (int)((((long long)(i + 1) * sw) + dw - 1) / dw);
                uint32_t sum[4] = {0, 0, 0, 0};
                for (int y = j0; y < j1; y++)
                    for (int x = i0; x < i1; x++) {
                        uint32_t t = src[(size_t)y * sw + x];
                        sum[0] += t & 255u; sum[1] += (t >> 8) & 255u; sum[2] += (t >> 16) & 255u; sum[3] += t >> 24;
                    }
                uint32_t cnt = (uint32_t)((j1 - j0) * (i1 - i0));
                uint32_t o = 0;
                for (int c = 0; c < 4; c++) o |= ((sum[c] + cnt / 2) / cnt) << (8 * c);
                nxt[(size_t)j * dw + i] = o;
            }
        }
        levels.push_back(nxt);
        sw = dw; sh = dh;
    }
}

extern "C" int mwb_set_texture(mwb_handle *h, int tex_id, int width, int height, const uint8_t *rgb) {
    if (!h || !rgb) return set_err(MWB_EINVAL, "mwb_set_texture: null argument");
    if (tex_id < 0 || tex_id >= MWB_MAX_TEX || width <= 0 || height <= 0 || width > 4096 || height > 4096)
        return set_err(MWB_EINVAL, "mwb_set_texture: bad texture id or size");
    build_mips(rgb, width, height, h->tex_levels[tex_id]);
    if ((int)h->tex_levels[tex_id].size() > MWB_MAX_LEVELS) return set_err(MWB_EINVAL, "mwb_set_texture: too many mip levels");
    h->tex_w[tex_id] = width; h->tex_h[tex_id] = height;
    h->textures_dirty = true;
    return MWB_OK;
}

static int upload_textures(mwb_handle *h) {
    USE_DEVICE(h->cfg.device);
    const int n_tex = h->dev.n_tex;
    for (int i = 0; i < n_tex; i++)
        if (h->tex_w[i] == 0)
            return set_err(MWB_ESTATE, "render requested before the task's " + std::to_string(n_tex) + " textures were set (mwb_set_texture)");
    std::vector<uint32_t> all;
    MwbTexDesc desc[MWB_MAX_TEX];
    memset(desc, 0, sizeof(desc));
    for (int i = 0; i < n_tex; i++) {
        desc[i].w = h->tex_w[i]; desc[i].h = h->tex_h[i]; desc[i].n_levels = (int)h->tex_levels[i].size();
        desc[i].sc_s = (float)(512.0 / h->tex_w[i]); desc[i].sc_t = (float)(512.0 / h->tex_h[i]);
        for (size_t l = 0; l < h->tex_levels[i].size(); l++) {
            desc[i].level_off[l] = (uint32_t)all.size();
            all.insert(all.end(), h->tex_levels[i][l].begin(), h->tex_levels[i][l].end());
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    if (h->texels_dev) { hipFree(h->texels_dev); h->texels_dev = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->texels_dev, all.size() * 4));
    HIP_TRY(hipMemcpy(h->texels_dev, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->tex_desc_dev, desc, sizeof(desc), hipMemcpyHostToDevice));
    h->dev.texels = h->texels_dev;
    h->textures_dirty = false; h->have_textures = true;
    return MWB_OK;
}

// -------------------------------------------------------------------------------------- meshes
extern "C" int mwb_set_mesh(mwb_handle *h, int geom, int n_tris, const float *verts, const float *norms, const float *texcs, int tex_slot,
                            const float *min_coords, const float *max_coords, int n_nodes, int n_orders, const float *nodes, const int32_t *perm) {
    if (!h || !verts || !norms || !texcs || !min_coords || !max_coords || !nodes || !perm) return set_err(MWB_EINVAL, "mwb_set_mesh: null argument");
    if (geom < 0 || geom >= MWB_NUM_MESHES || n_tris <= 0 || n_tris >= (1 << 24) || n_nodes <= 0 || tex_slot >= MWB_MAX_TEX || (n_orders != 1 && n_orders != 8))
        return set_err(MWB_EINVAL, "mwb_set_mesh: bad geometry id / sizes");
    for (int ord = 0; ord < n_orders; ord++) {   // the hierarchy is walked by a kernel: every link must stay inside the arrays and move forward (the walk terminates)
        std::vector<char> seen((size_t)n_tris, 0);
        const float *on = nodes + (size_t)ord * n_nodes * 8;
        for (int i = 0; i < n_nodes; i++) {
            int32_t skip, fc;
            memcpy(&skip, on + (size_t)i * 8 + 3, 4); memcpy(&fc, on + (size_t)i * 8 + 7, 4);
            const int cnt = (int)((uint32_t)fc >> 24), first = (int)((uint32_t)fc & 0xFFFFFFu);
            if (skip <= i || skip > n_nodes || first + cnt > n_tris || (cnt == 0 && i + 1 >= n_nodes))
                return set_err(MWB_EINVAL, "mwb_set_mesh: malformed hierarchy");
            for (int k = 0; k < cnt; k++) seen[(size_t)first + k] = 1;
        }
        for (int i = 0; i < n_tris; i++)
            if (!seen[i] || perm[i] < 0 || perm[i] >= n_tris) return set_err(MWB_EINVAL, "mwb_set_mesh: the leaves must cover every triangle once");
    }
    mwb_handle::HostMesh &m = h->meshes[geom];
    m.set = true; m.n_tris = n_tris; m.n_nodes = n_nodes; m.n_orders = n_orders; m.tex_id = tex_slot;
    for (int k = 0; k < 3; k++) { m.min_c[k] = min_coords[k]; m.max_c[k] = max_coords[k]; }
    m.nodes.assign(nodes, nodes + (size_t)n_orders * n_nodes * 8);
    auto rec = [&](int tri, float *o) {   // v0.xyz e1.x | e1.yz e2.xy | e2.z idx 0 0 with e1 = v1 - v0, e2 = v2 - v0 in float32
        const float *v = verts + (size_t)tri * 9;
        const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = e1[0]; o[4] = e1[1]; o[5] = e1[2]; o[6] = e2[0]; o[7] = e2[1]; o[8] = e2[2];
        int32_t idx = tri; memcpy(o + 9, &idx, 4); o[10] = 0; o[11] = 0;
    };
    m.tris.resize((size_t)n_tris * 12); m.tris2.resize((size_t)n_tris * 12); m.shade.assign((size_t)n_tris * 16, 0.0f);
    for (int i = 0; i < n_tris; i++) {
        rec(perm[i], &m.tris[(size_t)i * 12]);
        rec(i, &m.tris2[(size_t)i * 12]);
        float *o = &m.shade[(size_t)i * 16];
        memcpy(o, norms + (size_t)i * 9, 36);
        memcpy(o + 9, texcs + (size_t)i * 6, 24);
    }
    h->meshes_dirty = true;
    return MWB_OK;
}

extern "C" int mwb_set_mesh_dims(mwb_handle *h, int geom, double height, double scale, double radius, int is_f32) {
    if (!h) return set_err(MWB_EINVAL, "mwb_set_mesh_dims: null handle");
    if (geom < 0 || geom >= MWB_NUM_MESHES || !(height > 0) || !(scale > 0) || !(radius > 0) || !std::isfinite(scale) || !std::isfinite(radius))
        return set_err(MWB_EINVAL, "mwb_set_mesh_dims: bad argument");
    MwbDev &d = h->dev;
    int k = 0;
    while (k < d.n_mesh_dims && !(d.mesh_dims[k].geom == geom && d.mesh_dims[k].height == height)) k++;
    if (k == MWB_MAX_MESH_DIMS) return set_err(MWB_EINVAL, "mwb_set_mesh_dims: table full");
    d.mesh_dims[k].geom = geom; d.mesh_dims[k].height = height; d.mesh_dims[k].scale = scale; d.mesh_dims[k].radius = radius; d.mesh_dims[k].is_f32 = is_f32 ? 1 : 0;
    if (k == d.n_mesh_dims) d.n_mesh_dims++;
    return MWB_OK;
}

// which geometries (and (geometry, height) pairs) the task builds
static const int *task_meshes(int task, int *n) {
    static const int pick[2] = {MWB_MESH_BALL, MWB_MESH_KEY}, health[1] = {MWB_MESH_MEDKIT}, three[3] = {MWB_MESH_DUCKIE, MWB_MESH_KEY, MWB_MESH_BALL},
                     sign[1] = {MWB_MESH_KEY}, side[2] = {MWB_MESH_BUILDING, MWB_MESH_CONE}, gap[1] = {MWB_MESH_BUILDING};
    switch (task) {
    case MWB_TASK_PICKUPOBJS: case MWB_TASK_ROOMOBJS: *n = 2; return pick;
    case MWB_TASK_COLLECTHEALTH: *n = 1; return health;
    case MWB_TASK_THREEROOMS: *n = 3; return three;
    case MWB_TASK_SIGN: *n = 1; return sign;
    case MWB_TASK_SIDEWALK: *n = 2; return side;
    case MWB_TASK_WALLGAP: *n = 1; return gap;
    }
    *n = 0; return nullptr;
}

static int upload_meshes(mwb_handle *h) {   // the caller holds the device guard
    int n_need = 0;
    const int *need = task_meshes(h->dev.task, &n_need);
    for (int k = 0; k < n_need; k++) {
        if (!h->meshes[need[k]].set) return set_err(MWB_ESTATE, "reset / render requested before the task's meshes were set (mwb_set_mesh)");
        bool dims = false;
        for (int q = 0; q < h->dev.n_mesh_dims; q++) dims = dims || h->dev.mesh_dims[q].geom == need[k];
        if (!dims) return set_err(MWB_ESTATE, "reset requested before the task's mesh dimensions were set (mwb_set_mesh_dims)");
    }
    std::vector<float> all;
    MwbMeshDesc desc[MWB_NUM_MESHES];
    memset(desc, 0, sizeof(desc));
    for (int g = 0; g < MWB_NUM_MESHES; g++) {
        const mwb_handle::HostMesh &m = h->meshes[g];
        if (!m.set) continue;
        desc[g].n_tris = m.n_tris; desc[g].n_nodes = m.n_nodes; desc[g].tex_id = m.tex_id; desc[g].n_orders = m.n_orders;
        for (int k = 0; k < 3; k++) { desc[g].min_c[k] = m.min_c[k]; desc[g].max_c[k] = m.max_c[k]; }
        desc[g].node_off = (uint32_t)(all.size() / 4); all.insert(all.end(), m.nodes.begin(), m.nodes.end());
        desc[g].tri_off = (uint32_t)(all.size() / 4); all.insert(all.end(), m.tris.begin(), m.tris.end());
        desc[g].tri2_off = (uint32_t)(all.size() / 4); all.insert(all.end(), m.tris2.begin(), m.tris2.end());
        desc[g].shade_off = (uint32_t)(all.size() / 4); all.insert(all.end(), m.shade.begin(), m.shade.end());
    }
    HIP_TRY(hipDeviceSynchronize());
    if (h->mesh_data_dev) { hipFree(h->mesh_data_dev); h->mesh_data_dev = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->mesh_data_dev, (all.size() ? all.size() : 4) * sizeof(float)));
    if (!all.empty()) HIP_TRY(hipMemcpy(h->mesh_data_dev, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->mesh_desc_dev, desc, sizeof(desc), hipMemcpyHostToDevice));
    h->dev.mesh_data = h->mesh_data_dev;
    h->meshes_dirty = false;
    return MWB_OK;
}

// ---------------------------------------------------------------------------------- simulation
extern "C" int mwb_seed(mwb_handle *h, const uint64_t *seeds) {
    if (!h || !seeds) return set_err(MWB_EINVAL, "mwb_seed: null argument");
    USE_DEVICE(h->cfg.device);
    std::vector<uint32_t> st((size_t)h->dev.N * MWB_MT_WORDS);
    for (int e = 0; e < h->dev.N; e++) {
        uint32_t key[2];
        int n = seed_to_key(seeds[e], key);
        mt_init_by_array(&st[(size_t)e * MWB_MT_WORDS], key, n);
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h->dev.rng, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    h->seeded = true;
    return MWB_OK;
}

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(MWB_EHIP, std::string(what) + " launch failed: " + hipGetErrorString(e));
    return MWB_OK;
}

#define EVN 7
static int timing_begin(mwb_handle *h, hipStream_t s) {
    h->timing_now = h->timing && (h->timing_tick++ % h->timing_period) == 0;
    if (!h->timing_now) return MWB_OK;
    if (h->ev_used + EVN > h->ev_pool.size() && h->ev_pool.size() >= (size_t)EVN * 65536)
        h->ev_used = 0;   // nobody read the samples for 65536 timed passes: start over instead of growing without bound
    if (h->ev_used + EVN > h->ev_pool.size()) {
        size_t old = h->ev_pool.size();
        h->ev_pool.resize(old + EVN * 256);
        for (size_t i = old; i < h->ev_pool.size(); i++) HIP_TRY(hipEventCreate(&h->ev_pool[i]));
    }
    h->ev = &h->ev_pool[h->ev_used];
    h->ev_used += EVN;
    HIP_TRY(hipEventRecord(h->ev[0], s));
    return MWB_OK;
}
#define TMARK(i) do { if (h->timing_now) HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)

static int render_tail(mwb_handle *h, int mode, hipStream_t s) {
    mwb_launch_prep(h->dev, mode, s);
    int rc = check_launch("prep_kernel"); if (rc) return rc;
    TMARK(3);
    mwb_launch_render(h->dev, mode, s);
    rc = check_launch("render_kernel"); if (rc) return rc;
    TMARK(4);
    return MWB_OK;
}

// Fused frame stack: move the window on for the pass that is about to render (0 = a step, 2 = a reset of everything), copying
// the history back to the front first when the window has reached the end of its planes.
static int stack_advance(mwb_handle *h, int after_reset, hipStream_t s) {
    if (!h->stack_fused) return MWB_OK;
    const int C = h->stack_n * 3, from = h->stack_pos;
    int pos = after_reset ? 0 : from + 3;
    if (!after_reset && pos + C > h->stack_planes) {
        mwb_launch_stack_slide(h->dev, h->stack, h->stack_n, h->stack_planes, h->stack_dtype, 0, from, 3, s);
        int rc = check_launch("stack_slide_kernel"); if (rc) return rc;
        pos = 0;
    }
    h->stack_pos = pos; h->dev.stk_pos = pos;
    return MWB_OK;
}

static int ensure_ready(mwb_handle *h, bool renders = true) {   // the caller holds the device guard
    if (!h->seeded) return set_err(MWB_ESTATE, "mwb_seed must be called before reset/step (the reference seeds from entropy; this library refuses to)");
    if (h->textures_dirty || !h->have_textures) { int rc = upload_textures(h); if (rc) return rc; }
    if (h->dev.ent_task && (h->meshes_dirty || !h->mesh_data_dev)) { int rc = upload_meshes(h); if (rc) return rc; }
    if (renders) h->have_obs = true;
    return MWB_OK;
}

extern "C" int mwb_reset(mwb_handle *h, const uint8_t *mask_dev, void *stream) {
    if (!h) return set_err(MWB_EINVAL, "null handle");
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    h->dev.step_pass = 0;
    rc = timing_begin(h, s); if (rc) return rc;
    rc = stack_advance(h, mask_dev == nullptr, s); if (rc) return rc;   // a full reset restarts the window; a partial one is a step for the others
    mwb_launch_mark_reset(h->dev, mask_dev, s);
    rc = check_launch("mark_reset_kernel"); if (rc) return rc;
    TMARK(1); TMARK(5);
    mwb_launch_reset(h->dev, MWB_RESET_MAX_BLOCKS, s);   // possibly every env: many blocks
    rc = check_launch("reset_kernel"); if (rc) return rc;
    TMARK(6); TMARK(2);
    rc = render_tail(h, 0, s); if (rc) return rc;
    mwb_launch_clear_list(h->dev, s);
    rc = check_launch("clear_list_kernel"); if (rc) return rc;
    // the first step's bulk render is dispatched by the frame costs this render just measured (not in env order)
    mwb_launch_order(h->dev, s);
    return check_launch("order_kernel");
}

static int step_impl(mwb_handle *h, const int32_t *actions_dev, const uint8_t *skip_mask_dev, void *stream);
extern "C" int mwb_step(mwb_handle *h, const int32_t *actions_dev, const uint8_t *skip_mask_dev, void *stream) {
    if (!h) return set_err(MWB_EINVAL, "null handle");
    h->dev.act_stride = 1;
    return step_impl(h, actions_dev, skip_mask_dev, stream);
}
extern "C" int mwb_step_i64(mwb_handle *h, const int64_t *actions_dev, const uint8_t *skip_mask_dev, void *stream) {
    if (!h) return set_err(MWB_EINVAL, "null handle");
    h->dev.act_stride = 2;   // little-endian: the low word of each int64 (actions are small non-negative integers)
    return step_impl(h, (const int32_t *)actions_dev, skip_mask_dev, stream);
}
static int step_impl(mwb_handle *h, const int32_t *actions_dev, const uint8_t *skip_mask_dev, void *stream) {
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h); if (rc) return rc;
    if (!actions_dev) return set_err(MWB_EINVAL, "mwb_step: null actions");
    hipStream_t s = (hipStream_t)stream;
    h->dev.step_pass = 1;   // the frames of this pass are the step's own (entity tasks: what a task rule removed is still drawn)
    rc = timing_begin(h, s); if (rc) return rc;
    rc = stack_advance(h, 0, s); if (rc) return rc;
    mwb_launch_step(h->dev, actions_dev, skip_mask_dev, s);
    rc = check_launch("step_kernel"); if (rc) return rc;
    TMARK(1);
    if (!h->overlap_reset) {
        TMARK(5);
        mwb_launch_reset(h->dev, 1024, s);
        rc = check_launch("reset_kernel"); if (rc) return rc;
        TMARK(6); TMARK(2);
        rc = render_tail(h, 0, s); if (rc) return rc;
        mwb_launch_clear_list(h->dev, s);
        return check_launch("clear_list_kernel");
    }
    // fork: the few envs that ended are regenerated, prepared and rendered on the side stream while the
    // caller's stream renders everybody else; join before returning control of the outputs
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if (h->timing_now) HIP_TRY(hipEventRecord(h->ev[5], h->side));
    mwb_launch_reset(h->dev, 1024, h->side);   // a handful of envs end per step - or all of them (mass time-out); started ahead of the bulk render
    rc = check_launch("reset_kernel"); if (rc) return rc;
    if (h->timing_now) HIP_TRY(hipEventRecord(h->ev[6], h->side));
    mwb_launch_render(h->dev, 1, h->side);   // reset_kernel has prepared the frame constants of the envs it regenerated
    rc = check_launch("render_kernel"); if (rc) return rc;
    mwb_launch_clear_list(h->dev, h->side);   // the list is consumed; off the critical path
    // also off the critical path: the next step's dispatch order, from the frame costs measured so far
    mwb_launch_order(h->dev, h->side);
    rc = check_launch("order_kernel"); if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_join, h->side));
    TMARK(2);
    rc = render_tail(h, 2, s); if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    return MWB_OK;   // the map filled beside this step is adopted by the next step_kernel (device-side flip: graph-capturable)
}

extern "C" int mwb_render(mwb_handle *h, void *stream) {
    if (!h) return set_err(MWB_EINVAL, "null handle");
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    h->dev.step_pass = 0;   // the current state (render_obs() called again shows what the task rule left)
    rc = timing_begin(h, s); if (rc) return rc;
    TMARK(1); TMARK(5); TMARK(6); TMARK(2);
    return render_tail(h, 0, s);
}

extern "C" int mwb_render_top_view(mwb_handle *h, uint8_t *out_dev, int width, int height, void *stream) {
    if (!h || !out_dev) return set_err(MWB_EINVAL, "mwb_render_top_view: null argument");
    if (width < 1 || height < 1 || width > 4096 || height > 4096) return set_err(MWB_EINVAL, "mwb_render_top_view: bad frame size");
    if (h->dev.ent_task) return set_err(MWB_EINVAL, "mwb_render_top_view: not available for the tasks with mesh entities");
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h, false); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    mwb_launch_prep(h->dev, 0, s);   // the state may have been set from outside since the last render
    rc = check_launch("prep_kernel"); if (rc) return rc;
    mwb_launch_top_view(h->dev, out_dev, width, height, s);
    return check_launch("top_view_kernel");
}

extern "C" int mwb_render_view(mwb_handle *h, uint8_t *out_dev, float *depth_dev, int width, int height, void *stream) {
    if (!h || !out_dev) return set_err(MWB_EINVAL, "mwb_render_view: null argument");
    if (width < 1 || height < 1 || width > 4096 || height > 4096) return set_err(MWB_EINVAL, "mwb_render_view: bad frame size");
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h, false); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t words = (size_t)h->dev.N * h->dev.frame_words;
    if (!h->view_frame) {   // frame constants of the other view size, kept apart from the observation's
        rc = dev_alloc(h, &h->view_frame, words); if (rc) return rc;
    }
    MwbDev v = h->dev;
    v.W = width; v.H = height; v.layout = MWB_LAYOUT_HWC; v.obs = out_dev; v.depth = depth_dev; v.want_depth = depth_dev ? 1 : 0;
    v.frame = h->view_frame; v.stk = nullptr; v.step_pass = 0; v.wg_ts = nullptr;
    mwb_launch_prep(v, 0, s);
    rc = check_launch("prep_kernel"); if (rc) return rc;
    const int lrc = mwb_launch_render_view(v, s);
    if (lrc) return set_err(lrc == -1 ? MWB_EINVAL : MWB_EHIP, lrc == -1 ? "mwb_render_view: the world's room table does not fit LDS beside a tile" : "mwb_render_view: hipFuncSetAttribute failed");
    return check_launch("render_view_kernel");
}

extern "C" int mwb_visible_ents(mwb_handle *h, uint32_t *mask_dev, void *stream) {
    if (!h || !mask_dev) return set_err(MWB_EINVAL, "mwb_visible_ents: null argument");
    USE_DEVICE(h->cfg.device);
    int rc = ensure_ready(h, false); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    mwb_launch_prep(h->dev, 0, s);   // the state may have been set from outside since the last render
    rc = check_launch("prep_kernel"); if (rc) return rc;
    mwb_launch_visible(h->dev, mask_dev, s);
    return check_launch("visible_kernel");
}

extern "C" int mwb_get_outputs(mwb_handle *h, mwb_outputs *out) {
    if (!h || !out) return set_err(MWB_EINVAL, "mwb_get_outputs: null argument");
    const MwbDev &d = h->dev;
    out->obs = d.obs; out->depth = d.depth; out->reward = d.reward; out->reward64 = d.reward64; out->done = d.done;
    out->ep_steps = d.ep_steps;
    out->obs_bytes = (size_t)d.N * d.W * d.H * 3;
    out->depth_bytes = d.want_depth ? (size_t)d.N * d.W * d.H * 4 : 0;
    out->stack = h->stack; out->stack_bytes = h->stack_bytes;
    out->feature = d.feature; out->goal_pos = d.goal_pos;
    out->pack = h->pack; out->pack_bytes = h->pack_bytes;
    return MWB_OK;
}

extern "C" int mwb_check(mwb_handle *h) {
    if (!h) return set_err(MWB_EINVAL, "mwb_check: null handle");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    int32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, h->dev.error_flag, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) return set_err(MWB_ESTATE, "world generation failed for env " + std::to_string(flag - 1) + " (portal / placement condition the reference asserts on)");
    return MWB_OK;
}

// ---------------------------------------------------------------------------------- frame stack
extern "C" int mwb_stack_enable(mwb_handle *h, int nstack, int dtype) {
    if (!h) return set_err(MWB_EINVAL, "mwb_stack_enable: null handle");
    const MwbDev &d = h->dev;
    const int fused = (dtype & MWB_STACK_FUSED) != 0;
    const int sliding = fused || (dtype & MWB_STACK_SLIDING) != 0;
    dtype &= ~(MWB_STACK_SLIDING | MWB_STACK_FUSED);
    if (nstack < 1 || nstack > 16 || (dtype != 0 && dtype != 1)) return set_err(MWB_EINVAL, "mwb_stack_enable: bad nstack / dtype");
    if (d.layout != MWB_LAYOUT_CWH) return set_err(MWB_EINVAL, "mwb_stack_enable: the frame stack is channel-first, create the handle with MWB_LAYOUT_CWH");
    if ((d.W * d.H) % 4) return set_err(MWB_EINVAL, "mwb_stack_enable: W*H must be a multiple of 4");
    if (h->stack) return set_err(MWB_ESTATE, "mwb_stack_enable: already enabled");
    // a fused window is filled by the render kernels as they produce frames: enabled after the first observation it would
    // lack the current frame until the next pass (the non-fused forms rebuild theirs from the observation buffer)
    if (fused && h->dev.tile_w > 0) return set_err(MWB_EINVAL, "mwb_stack_enable: MWB_STACK_FUSED is not available when observations are rendered in tiles (large frames, MWB_TILE): use MWB_STACK_SLIDING");
    if (fused && h->have_obs) return set_err(MWB_ESTATE, "mwb_stack_enable: MWB_STACK_FUSED must be enabled before the first mwb_reset / mwb_step / mwb_render");
    if (fused && dtype == 0 && (d.W * d.H) % 16) return set_err(MWB_EINVAL, "mwb_stack_enable: a fused uint8 stack needs W*H to be a multiple of 16");
    USE_DEVICE(h->cfg.device);
    const int planes = sliding ? nstack * 3 + 3 * MWB_STACK_SLACK_FRAMES : nstack * 3;
    size_t bytes = (size_t)d.N * planes * d.W * d.H * (dtype == 1 ? 4 : 1);
    uint8_t *p = nullptr;
    int rc = dev_alloc(h, &p, bytes);
    if (rc) return rc;
    h->stack = p; h->stack_n = nstack; h->stack_dtype = dtype; h->stack_bytes = bytes;
    h->stack_planes = sliding ? planes : 0; h->stack_pos = 0; h->stack_fused = fused;
    if (fused) {
        MwbDev &dd = h->dev;
        dd.stk = p; dd.stk_float = dtype == 1; dd.stk_C = nstack * 3; dd.stk_K = planes; dd.stk_pos = 0;
    }
    return MWB_OK;
}

extern "C" int mwb_stack_update(mwb_handle *h, int after_reset, void *stream) {
    if (!h || !h->stack) return set_err(MWB_ESTATE, "mwb_stack_update: call mwb_stack_enable first");
    USE_DEVICE(h->cfg.device);
    if (h->stack_fused) return MWB_OK;   // the step / reset that produced the observation has already put it into the window
    if (h->stack_planes) {   // sliding window: the host owns the window position (mwb_stack_window)
        const int C = h->stack_n * 3, from = h->stack_pos;
        int mode = 0, pos = from + 3;
        if (after_reset) { mode = 2; pos = 0; }
        else if (pos + C > h->stack_planes) { mode = 1; pos = 0; }
        mwb_launch_stack_slide(h->dev, h->stack, h->stack_n, h->stack_planes, h->stack_dtype, pos, from, mode, (hipStream_t)stream);
        h->stack_pos = pos;
        return check_launch("stack_slide_kernel");
    }
    mwb_launch_stack(h->dev, h->stack, h->stack_n, h->stack_dtype, after_reset, (hipStream_t)stream);
    return check_launch("stack_kernel");
}

extern "C" int mwb_stack_window(mwb_handle *h, int *first_plane, int *planes_per_env) {
    if (!h || !h->stack) return set_err(MWB_ESTATE, "mwb_stack_window: call mwb_stack_enable first");
    if (first_plane) *first_plane = h->stack_planes ? h->stack_pos : 0;
    if (planes_per_env) *planes_per_env = h->stack_planes ? h->stack_planes : h->stack_n * 3;
    return MWB_OK;
}

// -------------------------------------------------------------------------------- introspection
template <typename T>
static int fetch(T *dst, const T *src_dev, size_t first, size_t count, size_t stride) {
    if (!dst) return MWB_OK;
    HIP_TRY(hipMemcpy(dst, src_dev + first * stride, count * stride * sizeof(T), hipMemcpyDeviceToHost));
    return MWB_OK;
}

extern "C" int mwb_num_boxes(mwb_handle *h) { return h ? h->dev.n_boxes : 0; }
extern "C" int mwb_room_words(mwb_handle *h) { return h ? h->dev.room_words : 0; }

// box arrays live as [B][N] planes on the device and as [count][B](x width) rows in mwb_state
static int fetch_boxes(double *dst, const double *src_dev, size_t N, int B, size_t first, size_t count, size_t width) {
    if (!dst) return MWB_OK;
    std::vector<double> plane(count * width);
    for (int b = 0; b < B; b++) {
        HIP_TRY(hipMemcpy(plane.data(), src_dev + ((size_t)b * N + first) * width, count * width * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < count; i++)
            for (size_t k = 0; k < width; k++) dst[(i * B + b) * width + k] = plane[i * width + k];
    }
    return MWB_OK;
}
static int store_boxes(const double *src, double *dst_dev, size_t N, int B, size_t first, size_t count, size_t width) {
    if (!src) return MWB_OK;
    std::vector<double> plane(count * width);
    for (int b = 0; b < B; b++) {
        for (size_t i = 0; i < count; i++)
            for (size_t k = 0; k < width; k++) plane[i * width + k] = src[(i * B + b) * width + k];
        HIP_TRY(hipMemcpy(dst_dev + ((size_t)b * N + first) * width, plane.data(), count * width * sizeof(double), hipMemcpyHostToDevice));
    }
    return MWB_OK;
}
template <typename T>
static int store(const T *src, T *dst_dev, size_t first, size_t count, size_t stride) {
    if (!src) return MWB_OK;
    HIP_TRY(hipMemcpy(dst_dev + first * stride, src, count * stride * sizeof(T), hipMemcpyHostToDevice));
    return MWB_OK;
}

extern "C" int mwb_get_state(mwb_handle *h, int first, int count, mwb_state *o) {
    if (!h || !o) return set_err(MWB_EINVAL, "mwb_get_state: null argument");
    const MwbDev &d = h->dev;
    if (first < 0 || count < 0 || first + count > d.N) return set_err(MWB_EINVAL, "mwb_get_state: env range out of bounds");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)d.N;
    const int B = d.n_boxes;
    std::vector<double> a(count), b(count);
    int rc;
    if (o->agent_pos) {
        if ((rc = fetch(a.data(), d.agent_x, first, count, 1))) return rc;
        if ((rc = fetch(b.data(), d.agent_z, first, count, 1))) return rc;
        for (int i = 0; i < count; i++) { o->agent_pos[i * 3] = a[i]; o->agent_pos[i * 3 + 1] = 0.0; o->agent_pos[i * 3 + 2] = b[i]; }
    }
    if (o->box_pos) {
        std::vector<double> bx((size_t)count * B), bz((size_t)count * B), by((size_t)count * B);
        if ((rc = fetch_boxes(bx.data(), d.box_x, N, B, first, count, 1))) return rc;
        if ((rc = fetch_boxes(bz.data(), d.box_z, N, B, first, count, 1))) return rc;
        if ((rc = fetch_boxes(by.data(), d.box_y, N, B, first, count, 1))) return rc;
        for (size_t i = 0; i < (size_t)count * B; i++) { o->box_pos[i * 3] = bx[i]; o->box_pos[i * 3 + 1] = by[i]; o->box_pos[i * 3 + 2] = bz[i]; }
    }
    if ((rc = fetch(o->agent_dir, d.agent_dir, first, count, 1))) return rc;
    if ((rc = fetch_boxes(o->box_dir, d.box_dir, N, B, first, count, 1))) return rc;
    if ((rc = fetch_boxes(o->box_color, d.box_color, N, B, first, count, 3))) return rc;
    if ((rc = fetch_boxes(o->box_size, d.box_size, N, B, first, count, 1))) return rc;
    if ((rc = fetch(o->goal_dist, d.goal_dist, first, count, 1))) return rc;
    if ((rc = fetch(o->carrying, d.carrying, first, count, 1))) return rc;
    if (o->ent_meta || o->ent_radius || o->ent_height || o->ent_scale || o->ent_order || o->task_f || o->task_i || o->text_tex) {
        if (!d.ent_task) return set_err(MWB_EINVAL, "mwb_get_state: entity-list fields exist for the tasks >= MWB_TASK_PICKUPOBJS only");
        if (o->ent_meta) {
            std::vector<int32_t> plane(count);
            for (int b = 0; b < B; b++) {
                HIP_TRY(hipMemcpy(plane.data(), d.ent_meta + (size_t)b * N + first, count * sizeof(int32_t), hipMemcpyDeviceToHost));
                for (int i = 0; i < count; i++) o->ent_meta[(size_t)i * B + b] = plane[i];
            }
        }
        if ((rc = fetch_boxes(o->ent_radius, d.ent_radius, N, B, first, count, 1))) return rc;
        if ((rc = fetch_boxes(o->ent_height, d.ent_height, N, B, first, count, 1))) return rc;
        if ((rc = fetch_boxes(o->ent_scale, d.ent_scale, N, B, first, count, 1))) return rc;
        if (o->ent_order) {
            std::vector<uint8_t> ord((size_t)count * MWB_ORDER_STRIDE);
            std::vector<int32_t> no(count);
            HIP_TRY(hipMemcpy(ord.data(), d.ent_order + (size_t)first * MWB_ORDER_STRIDE, ord.size(), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(no.data(), d.n_order + first, count * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int i = 0; i < count; i++)
                for (int k = 0; k <= B; k++) {
                    const int v = ord[(size_t)i * MWB_ORDER_STRIDE + k];
                    o->ent_order[(size_t)i * (B + 1) + k] = k >= no[i] ? -1 : (v == MWB_ENT_AGENT ? -2 : v);
                }
        }
        if ((rc = fetch(o->task_f, d.task_f, first, count, 1))) return rc;
        if ((rc = fetch(o->task_i, d.task_i, first, count, 1))) return rc;
        if ((rc = fetch(o->text_tex, d.text_tex, first, count, 8))) return rc;
    }
    if ((rc = fetch(o->goal_idx, d.goal_idx, first, count, 1))) return rc;
    if ((rc = fetch(o->episode_count, d.episode_count, first, count, 1))) return rc;
    if ((rc = fetch(o->task_step_count, d.task_step_count, first, count, 1))) return rc;
    if ((rc = fetch(o->cam, d.cam, first, count, 4))) return rc;
    if ((rc = fetch(o->sky_color, d.sky_color, first, count, 3))) return rc;
    if ((rc = fetch(o->light_pos, d.light_pos, first, count, 3))) return rc;
    if ((rc = fetch(o->light_color, d.light_color, first, count, 3))) return rc;
    if ((rc = fetch(o->light_ambient, d.light_ambient, first, count, 3))) return rc;
    if ((rc = fetch(o->step_count, d.step_count, first, count, 1))) return rc;
    if ((rc = fetch(o->n_rooms, d.n_rooms, first, count, 1))) return rc;
    if ((rc = fetch(o->n_segs, d.n_segs, first, count, 1))) return rc;
    if (o->rng_pos || o->rng_keysum || o->rng_state) {
        std::vector<uint32_t> st((size_t)count * MWB_MT_WORDS);
        if ((rc = fetch(st.data(), d.rng, first, count, MWB_MT_WORDS))) return rc;
        for (int i = 0; i < count; i++) {
            uint64_t sum = 0;
            for (int k = 0; k < 624; k++) sum += st[(size_t)i * MWB_MT_WORDS + k];
            if (o->rng_keysum) o->rng_keysum[i] = (uint32_t)(sum & 0xFFFFFFFFu);
            if (o->rng_pos) o->rng_pos[i] = (int32_t)st[(size_t)i * MWB_MT_WORDS + 624];
        }
        if (o->rng_state) memcpy(o->rng_state, st.data(), st.size() * sizeof(uint32_t));
    }
    return MWB_OK;
}

extern "C" int mwb_set_state(mwb_handle *h, int first, int count, const mwb_state *in) {
    if (!h || !in) return set_err(MWB_EINVAL, "mwb_set_state: null argument");
    const MwbDev &d = h->dev;
    if (first < 0 || count < 0 || first + count > d.N) return set_err(MWB_EINVAL, "mwb_set_state: env range out of bounds");
    if (in->goal_idx)
        for (int i = 0; i < count; i++)
            if (in->goal_idx[i] < 0 || in->goal_idx[i] > 1) return set_err(MWB_EINVAL, "mwb_set_state: goal_idx must be 0 or 1");
    if (in->carrying)
        for (int i = 0; i < count; i++)
            if (in->carrying[i] < -1 || in->carrying[i] >= h->dev.n_boxes) return set_err(MWB_EINVAL, "mwb_set_state: carrying must be -1 or a box index");
    if (in->rng_state)
        for (int i = 0; i < count; i++)
            if (in->rng_state[(size_t)i * MWB_MT_WORDS + 624] > 624u) return set_err(MWB_EINVAL, "mwb_set_state: MT19937 position must be 0..624");
    {   // poses, sizes and camera parameters go straight into the step and render kernels: no NaN / inf, no empty boxes
        auto finite = [](const double *v, size_t n) { if (v) for (size_t i = 0; i < n; i++) if (!std::isfinite(v[i])) return false; return true; };
        const size_t c = (size_t)count, B_ = (size_t)h->dev.n_boxes;
        if (!finite(in->agent_pos, c * 3) || !finite(in->agent_dir, c) || !finite(in->box_pos, c * B_ * 3) || !finite(in->box_dir, c * B_) ||
            !finite(in->box_color, c * B_ * 3) || !finite(in->box_size, c * B_) || !finite(in->cam, c * 4) || !finite(in->sky_color, c * 3) ||
            !finite(in->light_pos, c * 3) || !finite(in->light_color, c * 3) || !finite(in->light_ambient, c * 3) || !finite(in->goal_dist, c))
            return set_err(MWB_EINVAL, "mwb_set_state: non-finite value");
        if (in->box_size)
            for (size_t i = 0; i < c * B_; i++)
                if (!(in->box_size[i] > 0)) return set_err(MWB_EINVAL, "mwb_set_state: box_size must be > 0");
    }
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)d.N;
    const int B = d.n_boxes;
    int rc;
    if (in->agent_pos) {
        std::vector<double> a(count), b(count);
        for (int i = 0; i < count; i++) { a[i] = in->agent_pos[i * 3]; b[i] = in->agent_pos[i * 3 + 2]; }
        if ((rc = store(a.data(), d.agent_x, first, count, 1))) return rc;
        if ((rc = store(b.data(), d.agent_z, first, count, 1))) return rc;
    }
    if (in->box_pos) {
        std::vector<double> bx((size_t)count * B), bz((size_t)count * B), by((size_t)count * B);
        for (size_t i = 0; i < (size_t)count * B; i++) { bx[i] = in->box_pos[i * 3]; by[i] = in->box_pos[i * 3 + 1]; bz[i] = in->box_pos[i * 3 + 2]; }
        if ((rc = store_boxes(bx.data(), d.box_x, N, B, first, count, 1))) return rc;
        if ((rc = store_boxes(by.data(), d.box_y, N, B, first, count, 1))) return rc;
        if ((rc = store_boxes(bz.data(), d.box_z, N, B, first, count, 1))) return rc;
    }
    if ((rc = store(in->agent_dir, d.agent_dir, first, count, 1))) return rc;
    if ((rc = store_boxes(in->box_dir, d.box_dir, N, B, first, count, 1))) return rc;
    if ((rc = store_boxes(in->box_color, d.box_color, N, B, first, count, 3))) return rc;
    if ((rc = store_boxes(in->box_size, d.box_size, N, B, first, count, 1))) return rc;
    if ((rc = store(in->goal_dist, d.goal_dist, first, count, 1))) return rc;
    if ((rc = store(in->carrying, d.carrying, first, count, 1))) return rc;
    if (in->ent_meta || in->ent_radius || in->ent_height || in->ent_scale || in->ent_order || in->task_f || in->task_i || in->text_tex) {
        if (!d.ent_task) return set_err(MWB_EINVAL, "mwb_set_state: entity-list fields exist for the tasks >= MWB_TASK_PICKUPOBJS only");
        if (in->ent_radius || in->ent_height || in->ent_scale || in->text_tex) return set_err(MWB_EINVAL, "mwb_set_state: ent_radius / ent_height / ent_scale / text_tex are read-only");
        if (in->ent_meta) {   // only the `alive` bit may differ (an entity PickupObjs removed): kinds and dimensions belong to the episode
            std::vector<int32_t> plane(count);
            for (int b = 0; b < B; b++) {
                HIP_TRY(hipMemcpy(plane.data(), d.ent_meta + (size_t)b * N + first, count * sizeof(int32_t), hipMemcpyDeviceToHost));
                for (int i = 0; i < count; i++) {
                    const int32_t v = in->ent_meta[(size_t)i * B + b];
                    if ((v & ~(1 << 9)) != (plane[i] & ~(1 << 9))) return set_err(MWB_EINVAL, "mwb_set_state: ent_meta may only change an entity's alive bit");
                    plane[i] = v;
                }
                HIP_TRY(hipMemcpy(d.ent_meta + (size_t)b * N + first, plane.data(), count * sizeof(int32_t), hipMemcpyHostToDevice));
            }
        }
        if (in->ent_order) {
            std::vector<uint8_t> ord((size_t)count * MWB_ORDER_STRIDE, (uint8_t)MWB_ENT_AGENT);
            std::vector<int32_t> no(count);
            for (int i = 0; i < count; i++) {
                int n = 0;
                uint32_t seen = 0;
                for (int k = 0; k <= B; k++) {
                    const int v = in->ent_order[(size_t)i * (B + 1) + k];
                    if (v == -1) break;
                    if (v != -2 && (v < 0 || v >= B)) return set_err(MWB_EINVAL, "mwb_set_state: ent_order entries are slots, -2 (the agent) or -1 (end)");
                    const uint32_t bit = 1u << (v == -2 ? 31 : v);
                    if (seen & bit) return set_err(MWB_EINVAL, "mwb_set_state: ent_order lists an entity twice");
                    seen |= bit;
                    ord[(size_t)i * MWB_ORDER_STRIDE + n++] = (uint8_t)(v == -2 ? MWB_ENT_AGENT : v);
                }
                no[i] = n;
            }
            HIP_TRY(hipMemcpy(d.ent_order + (size_t)first * MWB_ORDER_STRIDE, ord.data(), ord.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d.n_order + first, no.data(), count * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if ((rc = store(in->task_f, d.task_f, first, count, 1))) return rc;
        if ((rc = store(in->task_i, d.task_i, first, count, 1))) return rc;
    }
    if ((rc = store(in->goal_idx, d.goal_idx, first, count, 1))) return rc;
    if ((rc = store(in->episode_count, d.episode_count, first, count, 1))) return rc;
    if ((rc = store(in->task_step_count, d.task_step_count, first, count, 1))) return rc;
    if ((rc = store(in->cam, d.cam, first, count, 4))) return rc;
    if ((rc = store(in->sky_color, d.sky_color, first, count, 3))) return rc;
    if ((rc = store(in->light_pos, d.light_pos, first, count, 3))) return rc;
    if ((rc = store(in->light_color, d.light_color, first, count, 3))) return rc;
    if ((rc = store(in->light_ambient, d.light_ambient, first, count, 3))) return rc;
    if ((rc = store(in->step_count, d.step_count, first, count, 1))) return rc;
    if ((rc = store(in->rng_state, d.rng, first, count, MWB_MT_WORDS))) return rc;
    return MWB_OK;
}

extern "C" int mwb_set_agent(mwb_handle *h, int first, int count, const double *pos_xz, const double *dir, const int32_t *step_count) {
    if (!h) return set_err(MWB_EINVAL, "mwb_set_agent: null handle");
    const MwbDev &d = h->dev;
    if (first < 0 || count < 0 || first + count > d.N) return set_err(MWB_EINVAL, "mwb_set_agent: env range out of bounds");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    if (pos_xz) {
        std::vector<double> a(count), b(count);
        for (int i = 0; i < count; i++) { a[i] = pos_xz[i * 2]; b[i] = pos_xz[i * 2 + 1]; }
        HIP_TRY(hipMemcpy(d.agent_x + first, a.data(), count * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.agent_z + first, b.data(), count * sizeof(double), hipMemcpyHostToDevice));
    }
    if (dir) HIP_TRY(hipMemcpy(d.agent_dir + first, dir, count * sizeof(double), hipMemcpyHostToDevice));
    if (step_count) HIP_TRY(hipMemcpy(d.step_count + first, step_count, count * sizeof(int32_t), hipMemcpyHostToDevice));
    return MWB_OK;
}

extern "C" int mwb_num_textures(mwb_handle *h) { return h ? h->dev.n_tex : 0; }

/* debugging aid (MWB_DEBUG bit 4 at mwb_create): start / end s_memrealtime ticks (100 MHz) of every workgroup of the last
 * bulk render launch; out: [2 * n] u64, returns n = workgroups or a negative code */
extern "C" int mwb_debug_wg_times(mwb_handle *h, unsigned long long *out, int max_wgs) {
    if (!h || !out) return set_err(MWB_EINVAL, "mwb_debug_wg_times: null argument");
    if (!h->dev.wg_ts) return set_err(MWB_ESTATE, "mwb_debug_wg_times: create the handle with MWB_DEBUG bit 4 set");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    int n = h->dev.N + h->dev.split_envs;
    if (n > max_wgs) n = max_wgs;
    HIP_TRY(hipMemcpy(out, h->dev.wg_ts, (size_t)n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}

/* timing experiments (MWB_EXP bit 2 at mwb_create): the entity render kernel's mesh-walk counters since the last call with
 * reset != 0: sample rays entering the walk, walks started, node visits, triangle tests, wave loop iterations, wave calls */
extern "C" int mwb_debug_counters(mwb_handle *h, unsigned long long *out8, int reset) {
    if (!h || !out8) return set_err(MWB_EINVAL, "mwb_debug_counters: null argument");
    if (!h->dev.dbg_counters) return set_err(MWB_ESTATE, "mwb_debug_counters: create the handle with MWB_EXP bit 2 set");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, h->dev.dbg_counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(h->dev.dbg_counters, 0, 8 * sizeof(unsigned long long)));
    return MWB_OK;
}

extern "C" int mwb_set_domain_rand(mwb_handle *h, int domain_rand) {
    if (!h) return set_err(MWB_EINVAL, "mwb_set_domain_rand: null handle");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());   // kernels in flight took the struct by value; order the change after them
    h->dev.domain_rand = domain_rand ? 1 : 0;
    h->cfg.domain_rand = h->dev.domain_rand;
    return MWB_OK;
}

extern "C" int mwb_set_task_state(mwb_handle *h, int first, int count, const int64_t *episode_count, const int64_t *task_step_count,
                                  const int32_t *goal_idx) {
    if (!h) return set_err(MWB_EINVAL, "mwb_set_task_state: null handle");
    const MwbDev &d = h->dev;
    if (first < 0 || count < 0 || first + count > d.N) return set_err(MWB_EINVAL, "mwb_set_task_state: env range out of bounds");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    if (episode_count) HIP_TRY(hipMemcpy(d.episode_count + first, episode_count, count * sizeof(int64_t), hipMemcpyHostToDevice));
    if (task_step_count) HIP_TRY(hipMemcpy(d.task_step_count + first, task_step_count, count * sizeof(int64_t), hipMemcpyHostToDevice));
    if (goal_idx) {
        for (int i = 0; i < count; i++)
            if (goal_idx[i] < 0 || goal_idx[i] > 1) return set_err(MWB_EINVAL, "mwb_set_task_state: goal_idx must be 0 or 1");
        HIP_TRY(hipMemcpy(d.goal_idx + first, goal_idx, count * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return MWB_OK;
}

extern "C" int mwb_intersect(mwb_handle *h, int env, int ent, double x, double z, double radius, int *result) {
    if (!h || !result) return set_err(MWB_EINVAL, "mwb_intersect: null argument");
    if (env < 0 || env >= h->dev.N) return set_err(MWB_EINVAL, "mwb_intersect: env out of range");
    if (ent < -1 || ent > h->dev.n_boxes) return set_err(MWB_EINVAL, "mwb_intersect: ent must be -1 .. number of boxes (= the agent)");
    USE_DEVICE(h->cfg.device);
    mwb_launch_intersect(h->dev, env, ent, x, z, radius, h->scratch_int_dev, 0);
    int rc = check_launch("intersect_kernel"); if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(result, h->scratch_int_dev, sizeof(int), hipMemcpyDeviceToHost));
    return MWB_OK;
}

extern "C" int mwb_get_geometry(mwb_handle *h, int env, float *rooms, int max_rooms, double *segs, int max_segs, int *n_rooms, int *n_segs) {
    if (!h || !n_rooms || !n_segs) return set_err(MWB_EINVAL, "mwb_get_geometry: null argument");
    const MwbDev &d = h->dev;
    if (env < 0 || env >= d.N) return set_err(MWB_EINVAL, "mwb_get_geometry: env out of range");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(n_rooms, d.n_rrooms + env, sizeof(int), hipMemcpyDeviceToHost));   // the table's records, cut rooms included
    HIP_TRY(hipMemcpy(n_segs, d.n_segs + env, sizeof(int), hipMemcpyDeviceToHost));
    if (rooms) {
        int n = *n_rooms < max_rooms ? *n_rooms : max_rooms;
        if (n > 0) HIP_TRY(hipMemcpy(rooms, d.rooms + (size_t)env * d.R_max * d.room_words, (size_t)n * d.room_words * 4, hipMemcpyDeviceToHost));
    }
    if (segs) {   // device layout is transposed: element (i, c) of env e at segs[(i*4 + c) * N + e]
        int n = *n_segs < max_segs ? *n_segs : max_segs;
        if (n > 0)
            HIP_TRY(hipMemcpy2D(segs, sizeof(double), d.segs + env, (size_t)d.N * sizeof(double), sizeof(double), (size_t)n * 4,
                                hipMemcpyDeviceToHost));
    }
    return MWB_OK;
}

// -------------------------------------------------------------------------------------- timing
extern "C" int mwb_timing_enable(mwb_handle *h, int enable) {
    if (!h) return set_err(MWB_EINVAL, "mwb_timing_enable: null handle");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    h->timing = enable != 0;
    h->timing_period = enable > 1 ? enable : 1;
    h->timing_tick = 0;
    h->ev_used = 0;
    if (h->timing && h->ev_pool.empty()) {   // the first block of events is created here, not inside somebody's timed region
        h->ev_pool.resize((size_t)EVN * 256);
        for (size_t i = 0; i < h->ev_pool.size(); i++) HIP_TRY(hipEventCreate(&h->ev_pool[i]));
    }
    return MWB_OK;
}

extern "C" int mwb_timing_read(mwb_handle *h, double *ms_step, double *ms_reset, double *ms_prep, double *ms_render, int *n) {
    if (!h) return set_err(MWB_EINVAL, "mwb_timing_read: null handle");
    USE_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    double acc[4] = {0, 0, 0, 0};
    int passes = (int)(h->ev_used / EVN);
    static const int A_[4] = {0, 5, 2, 3}, B_[4] = {1, 6, 3, 4};   // step, reset (its own stream), prep, render
    for (int p = 0; p < passes; p++)
        for (int i = 0; i < 4; i++) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[p * EVN + A_[i]], h->ev_pool[p * EVN + B_[i]]));
            acc[i] += ms;
        }
    int cnt = passes > 0 ? passes : 1;
    if (ms_step) *ms_step = acc[0] / cnt;
    if (ms_reset) *ms_reset = acc[1] / cnt;
    if (ms_prep) *ms_prep = acc[2] / cnt;
    if (ms_render) *ms_render = acc[3] / cnt;
    if (n) *n = passes;
    h->ev_used = 0;
    return MWB_OK;
}
