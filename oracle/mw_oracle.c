/* mw_oracle.c - CPU restatement of the gym-miniworld hot path (see mw_oracle.h for status).
 *
 * TEST INFRASTRUCTURE ONLY - never linked into or called by the product.
 *
 * State half: float64, one env at a time, written to follow the reference's operation
 * order so results are bit-identical to the reference run in this container
 * (tests/golden/state_*.npz).  Citations are file:line under /root/reference/.
 * Render half: float32 restatement of the fixed-function GL pipeline as configured by
 * miniworld.py:1014-1085,1160-1220 and opengl.py:85-103,283-371 - the frozen choices are
 * listed in DESIGN.md ("render spec") and in the comments below.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: a fused multiply-add would change last bits.
 */
#include "mw_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ===================================================================== MT19937 / numpy */
/* numpy.random.RandomState (legacy) draw recipes; reference random.py:4-65 calls
 * randint / uniform / choice on gym's RandomState. */
typedef struct {
    uint32_t key[624];
    int pos;
} MT;

static void mt_init_genrand(MT *m, uint32_t s) {
    m->key[0] = s;
    for (int i = 1; i < 624; i++) m->key[i] = 1812433253u * (m->key[i - 1] ^ (m->key[i - 1] >> 30)) + (uint32_t)i;
    m->pos = 624;
}

static void mt_init_by_array(MT *m, const uint32_t *init_key, int key_length) {
    int i = 1, j = 0, k;
    mt_init_genrand(m, 19650218u);
    k = (624 > key_length ? 624 : key_length);
    for (; k; k--) {
        m->key[i] = (m->key[i] ^ ((m->key[i - 1] ^ (m->key[i - 1] >> 30)) * 1664525u)) + init_key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { m->key[0] = m->key[623]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (k = 623; k; k--) {
        m->key[i] = (m->key[i] ^ ((m->key[i - 1] ^ (m->key[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { m->key[0] = m->key[623]; i = 1; }
    }
    m->key[0] = 0x80000000u;
    m->pos = 624;
}

static void mt_twist(MT *m) {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    uint32_t y;
    int i;
    for (i = 0; i < 624 - 397; i++) {
        y = (m->key[i] & UPPER) | (m->key[i + 1] & LOWER);
        m->key[i] = m->key[i + 397] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
    }
    for (; i < 623; i++) {
        y = (m->key[i] & UPPER) | (m->key[i + 1] & LOWER);
        m->key[i] = m->key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
    }
    y = (m->key[623] & UPPER) | (m->key[0] & LOWER);
    m->key[623] = m->key[396] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
    m->pos = 0;
}

static uint32_t mt_next32(MT *m) {
    if (m->pos == 624) mt_twist(m);
    uint32_t y = m->key[m->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_double(MT *m) {
    uint32_t a = mt_next32(m) >> 5, b = mt_next32(m) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* RandomState.uniform scalar/array element: loc + scale*u with scale = high - low */
static double rs_uniform(MT *m, double lo, double hi) {
    double scale = hi - lo;
    return lo + scale * mt_double(m);
}

/* RandomState.randint(low, high) int64 default, masked rejection on 32-bit words */
static long rs_randint(MT *m, long lo, long hi) {
    uint64_t rng = (uint64_t)(hi - 1 - lo);
    if (rng == 0) return lo;
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint32_t v;
    do { v = mt_next32(m) & (uint32_t)mask; } while (v > rng);
    return lo + (long)v;
}

/* RandomState.choice(n, p): one double, cdf.searchsorted(u, side='right') */
static int rs_choice_cdf(MT *m, const double *cdf, int n) {
    double u = mt_double(m);
    int lo = 0, hi = n;
    while (lo < hi) { /* first index with cdf[idx] > u */
        int mid = (lo + hi) / 2;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* numpy add.reduce over a contiguous double vector (pairwise summation, numpy
 * core/src/umath/loops_utils.h.src pairwise_sum): used by np.sum in miniworld.py:998 */
static double np_pairwise_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* ============================================================================ world */
typedef struct { double start, end, min_y, max_y; } Portal;

typedef struct {
    double verts[4][3];
    double norm[3];
    float texcs[4][2];
} Quad;

typedef struct {
    int n_edges;               /* num_walls: 4, or 3 for YMaze's triangular hub */
    double outline[4][3];      /* miniworld.py:90-97, y = 0 */
    double min_x, max_x, min_z, max_z, mid_x, mid_z, area;
    double edge_dirs[4][3], edge_norms[4][3];
    double wall_height;
    int no_ceiling;
    int wall_tex_name, floor_tex_name, ceil_tex_name; /* texture family ids */
    int n_portals[4];
    Portal portals[4][MWO_MAX_PORTALS];
    int nbr[4]; /* room behind the (single) portal of each edge, -1 none: recorded by connect_rooms for the polygon renderer */
    /* static data (miniworld.py:234-388) */
    int wall_tex, floor_tex, ceil_tex; /* concrete texture ids */
    int n_quads, n_segs;
    Quad *quads;
    double (*segs)[2][3];
    double floor_texcs[4][2], ceil_texcs[4][2];
} Room;

/* one entity of self.entities.  kind / geom / flags: reference entity.py; rad_f32: the radius is a numpy float32 scalar
 * (MeshEnt under NumPy >= 2, entity.py:118-127) - sums of it with Python floats are then evaluated in float32 */
typedef struct {
    double pos[3], dir, radius, height;
    int kind, geom, is_static, alive, rad_f32;
    double scale;
    double frame_w, frame_h, frame_d;   /* ImageFrame / TextFrame: width, height, depth */
    int tex[8], n_chars;                /* ImageFrame: tex[0]; TextFrame: one texture per character, -1 = space */
} Ent;
#define AGENT_SLOT (-2)

/* texture families (name -> list of <name>_<i>.png, opengl.py:40-69) */
enum { TEXF_FLOOR_TILES_BW = 0, TEXF_CONCRETE, TEXF_CONCRETE_TILES, TEXF_BRICK_WALL,
       TEXF_CARDBOARD, TEXF_WOOD, TEXF_WOOD_PLANKS, TEXF_DRYWALL, TEXF_STUCCO, TEXF_CEILING_TILES,
       TEXF_ASPHALT, TEXF_SLIME, TEXF_CINDER_BLOCKS, TEXF_LOGO_MILA, N_TEXF };
static const int TEXF_FIRST[N_TEXF] = {0, 1, 5, 6, 7, 11, 13, 14, 15, 16, 17, 18, 19, 20};
static const int TEXF_COUNT[N_TEXF] = {1, 4, 1, 1, 4, 2, 1, 1, 1, 1, 1, 1, 1, 1};
/* texture slots 21-24: the images of the textured meshes (medkit, duckie, building, cone); 25 + 9 c + v: variant v + 1 of the
 * c-th character of "BLUERDGN" (textures/chars/ch_0x<ord>_<v + 1>.png, entity.py:268-278) */
#define TEX_MESH0 21
#define TEX_CHAR0 25
static const char SIGN_CHARS[] = "BLUERDGN";

typedef struct {
    int n_tris, tex_id;
    float *verts, *norms, *texcs;   /* [n][3][3], [n][3][3], [n][3][2] as objmesh.py builds them */
    float *e1, *e2;                 /* [n][3]: v1 - v0, v2 - v0 in float32 (the ray / triangle test's inputs) */
    float min_c[3], max_c[3];
} Mesh;
static Mesh g_mesh[MWO_NMESH];
typedef struct { int geom; double height, scale, radius; int is_f32; } MeshDims;
static MeshDims g_mesh_dims[32];
static int g_n_mesh_dims;

typedef struct {
    int w, h, n_levels;
    uint8_t *data;
    size_t level_off[MWO_MAX_LEVELS];
    int lw[MWO_MAX_LEVELS], lh[MWO_MAX_LEVELS];
} Tex;
static Tex g_tex[MWO_MAX_TEX];

typedef struct { double def[3], lo[3], hi[3]; int n, is_int; } Param;

struct MwoEnv {
    int task;
    double task_args[4];
    int max_episode_steps, domain_rand;
    Param params[MWO_NPARAM];
    MT rng;
    int step_count;
    int n_rooms;
    Room rooms[MWO_MAX_ROOMS];
    int static_done;
    int n_segs;
    double (*wall_segs)[2][3];
    double room_probs[MWO_MAX_ROOMS], room_cdf[MWO_MAX_ROOMS];
    Ent boxes[MWO_MAX_BOXES], agent;   /* "boxes" = every entity but the agent, by slot (its position in the episode's first list) */
    int n_boxes; /* 1; 2 for the two-box T-maze (red, blue) and SimToRealPush (red, yellow); 6 for PutNext (COLOR_NAMES order); ... */
    int order[MWO_MAX_BOXES + 1], n_order; /* self.entities now, as slots (AGENT_SLOT = the agent): PickupObjs removes entries,
                                              CollectHealth moves a respawned kit to the end (collecthealth.py:56-57) */
    double health; int num_picked;          /* CollectHealth.health, PickupObjs.num_picked_up */
    /* what the reference's step() rendered: the entities as they were after the base step, before the task rule changed the list */
    double frame_pos[MWO_MAX_BOXES][3], frame_dir[MWO_MAX_BOXES];
    int frame_alive[MWO_MAX_BOXES], render_step_frame;
    double box_s[MWO_MAX_BOXES]; /* edge length of each box (Box(size=s), entity.py:366-378) */
    double box_colors[MWO_MAX_BOXES][3]; /* Box.color_vec after randomize (entity.py:381-383) */
    int box_base[MWO_MAX_BOXES]; /* index into COLORS (entity.py:8-15) */
    int carrying; /* index of the box agent.carrying refers to, or -1 (miniworld.py:682-702) */
    double goal_dist; /* SimToRealPush */
    /* T-maze family (envs/tmaze.py): goal alternation state */
    long long episode_count, task_step_count;
    int goal_idx;
    double feature[2];
    double cam_height, cam_fwd_disp, cam_pitch, cam_fov_y;
    double sky_color[3], light_pos[3], light_color[3], light_ambient[3];
    double max_forward_step;
};

static void fail(const char *msg) {
    fprintf(stderr, "mw_oracle: %s\n", msg);
    abort();
}

int mwo_set_mesh(int geom, int n_tris, const float *verts, const float *norms, const float *texcs, int tex_id,
                 const float *min_coords, const float *max_coords) {
    if (geom < 0 || geom >= MWO_NMESH || n_tris <= 0) return -1;
    Mesh *m = &g_mesh[geom];
    free(m->verts); free(m->norms); free(m->texcs); free(m->e1); free(m->e2);
    m->n_tris = n_tris; m->tex_id = tex_id;
    m->verts = malloc(sizeof(float) * 9 * n_tris); m->norms = malloc(sizeof(float) * 9 * n_tris); m->texcs = malloc(sizeof(float) * 6 * n_tris);
    m->e1 = malloc(sizeof(float) * 3 * n_tris); m->e2 = malloc(sizeof(float) * 3 * n_tris);
    memcpy(m->verts, verts, sizeof(float) * 9 * n_tris); memcpy(m->norms, norms, sizeof(float) * 9 * n_tris);
    memcpy(m->texcs, texcs, sizeof(float) * 6 * n_tris);
    for (int i = 0; i < n_tris; i++)
        for (int k = 0; k < 3; k++) { m->e1[i * 3 + k] = verts[i * 9 + 3 + k] - verts[i * 9 + k]; m->e2[i * 3 + k] = verts[i * 9 + 6 + k] - verts[i * 9 + k]; }
    for (int k = 0; k < 3; k++) { m->min_c[k] = min_coords[k]; m->max_c[k] = max_coords[k]; }
    return 0;
}
int mwo_set_mesh_dims(int geom, double height, double scale, double radius, int is_f32) {
    for (int i = 0; i < g_n_mesh_dims; i++)
        if (g_mesh_dims[i].geom == geom && g_mesh_dims[i].height == height) { g_mesh_dims[i] = (MeshDims){geom, height, scale, radius, is_f32}; return 0; }
    if (g_n_mesh_dims >= 32) return -1;
    g_mesh_dims[g_n_mesh_dims++] = (MeshDims){geom, height, scale, radius, is_f32};
    return 0;
}

int mwo_set_texture(int id, int w, int h, int n_levels, const uint8_t *rgba) {
    if (id < 0 || id >= MWO_MAX_TEX || n_levels > MWO_MAX_LEVELS) return -1;
    Tex *t = &g_tex[id];
    free(t->data);
    t->w = w; t->h = h; t->n_levels = n_levels;
    size_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < n_levels; l++) {
        t->level_off[l] = off; t->lw[l] = lw; t->lh[l] = lh;
        off += (size_t)lw * lh * 4;
        lw = lw > 1 ? lw / 2 : 1; lh = lh > 1 ? lh / 2 : 1;
    }
    t->data = (uint8_t *)malloc(off);
    memcpy(t->data, rgba, off);
    return 0;
}

static void default_params(Param *p) {
    /* params.py:110-123 */
    static const double T[MWO_NPARAM][10] = {
        /* n, def[3], lo[3], hi[3] */
        {3, 0.25, 0.82, 1, 0.1, 0.1, 0.1, 1.0, 1.0, 1.0},
        {3, 0, 2.5, 0, -40, 2.5, -40, 40, 5, 40},
        {3, 0.7, 0.7, 0.7, 0.45, 0.45, 0.45, 0.8, 0.8, 0.8},
        {3, 0.45, 0.45, 0.45, 0.35, 0.35, 0.35, 0.55, 0.55, 0.55},
        {3, 0, 0, 0, -0.2, -0.2, -0.2, 0.2, 0.2, 0.2},
        {1, 0.15, 0, 0, 0.12, 0, 0, 0.17, 0, 0},
        {1, 0, 0, 0, -0.05, 0, 0, 0.05, 0, 0},
        {1, 15, 0, 0, 10, 0, 0, 20, 0, 0},
        {1, 0.4, 0, 0, 0.38, 0, 0, 0.42, 0, 0},
        {1, 0, 0, 0, -5, 0, 0, 5, 0, 0},
        {1, 60, 0, 0, 55, 0, 0, 65, 0, 0},
        {1, 1.5, 0, 0, 1.45, 0, 0, 1.55, 0, 0},
        {1, 0, 0, 0, -0.05, 0, 0, 0.10, 0, 0},
    };
    for (int i = 0; i < MWO_NPARAM; i++) {
        p[i].n = (int)T[i][0]; p[i].is_int = 0;
        for (int k = 0; k < 3; k++) { p[i].def[k] = T[i][1 + k]; p[i].lo[k] = T[i][4 + k]; p[i].hi[k] = T[i][7 + k]; }
    }
}

/* params.py:81-99 DomainParams.sample: default when rng is None, else rng.float(min,max) */
static void sample_param(MwoEnv *e, int use_rng, int name, double *out) {
    Param *p = &e->params[name];
    for (int k = 0; k < p->n; k++)
        out[k] = use_rng ? rs_uniform(&e->rng, p->lo[k], p->hi[k]) : p->def[k];
}

MwoEnv *mwo_create(int task, const double *task_args, int max_episode_steps, int domain_rand, const double *params) {
    MwoEnv *e = (MwoEnv *)calloc(1, sizeof(MwoEnv));
    e->task = task;
    static const double dflt[MWO_NTASKS][4] = {{12, 0, 0, 0}, {10, 0, 0, 0}, {0, 0, 0, 0}, {8, 8, 3, 0}, {0, 0, 0, 0}, {0, 0, 0, 100}, {0, 0, 0, 0}, {0, 0, 0, 0}, {12, 0, 0, 0}, {0, 0, 0, 0},
                                               {12, 5, 0, 0}, {10, 0, 0, 0}, {16, 0, 0, 0}, {0, 0, 0, 0}, {10, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (task < 0 || task >= MWO_NTASKS) fail("unknown task");
    for (int i = 0; i < 4; i++) e->task_args[i] = task_args ? task_args[i] : dflt[task][i];
    if (task == MWO_PICKUPOBJS && e->task_args[1] == 0) e->task_args[1] = 5;
    e->n_boxes = (task == MWO_TMAZE_TWOBOX || task == MWO_SIM2REAL_PUSH) ? 2 : task == MWO_PUTNEXT ? 6 : 1;
    if (task == MWO_PICKUPOBJS) e->n_boxes = (int)e->task_args[1];
    else if (task == MWO_ROOMOBJS) e->n_boxes = 3;
    else if (task == MWO_COLLECTHEALTH) e->n_boxes = 18;
    else if (task == MWO_THREEROOMS) e->n_boxes = 6;
    else if (task == MWO_SIGN || task == MWO_SIDEWALK) e->n_boxes = 7;
    else if (task == MWO_WALLGAP) e->n_boxes = 2;
    if (e->n_boxes < 1 || e->n_boxes > MWO_MAX_BOXES) fail("too many entities");
    e->carrying = -1;
    if (max_episode_steps <= 0) {
        /* hallway.py:18, oneroom.py:14, fourrooms.py:15, maze.py:27 */
        if (task == MWO_HALLWAY) max_episode_steps = 250;
        else if (task == MWO_ONEROOM) max_episode_steps = 180;
        else if (task == MWO_FOURROOMS) max_episode_steps = 250;
        else if (task == MWO_TMAZE || task == MWO_TMAZE_TWOBOX) max_episode_steps = 280; /* tmaze.py:20,142 */
        else if (task == MWO_SIM2REAL_GOTO) max_episode_steps = 100; /* simtorealgoto.py:30 */
        else if (task == MWO_SIM2REAL_PUSH) max_episode_steps = 150; /* simtorealpush.py:29 */
        else if (task == MWO_PUTNEXT) max_episode_steps = 250; /* putnext.py:16 */
        else if (task == MWO_YMAZE) max_episode_steps = 280; /* ymaze.py:21 */
        else if (task == MWO_PICKUPOBJS) max_episode_steps = 400;      /* pickupobjs.py:19 */
        else if (task == MWO_ROOMOBJS) max_episode_steps = 2147483647; /* roomobjs.py:19: math.inf */
        else if (task == MWO_COLLECTHEALTH) max_episode_steps = 1000;  /* collecthealth.py:24 */
        else if (task == MWO_THREEROOMS) max_episode_steps = 400;      /* threerooms.py:14 */
        else if (task == MWO_SIGN) max_episode_steps = 20;             /* sign.py:41 */
        else if (task == MWO_SIDEWALK) max_episode_steps = 150;        /* sidewalk.py:15 */
        else if (task == MWO_WALLGAP) max_episode_steps = 300;         /* wallgap.py:14 */
        else max_episode_steps = (int)e->task_args[0] * (int)e->task_args[1] * 24;
    }
    e->max_episode_steps = max_episode_steps;
    e->domain_rand = domain_rand;
    default_params(e->params);
    if (params)
        for (int i = 0; i < MWO_NPARAM; i++)
            for (int k = 0; k < 3; k++) {
                e->params[i].def[k] = params[i * 9 + k];
                e->params[i].lo[k] = params[i * 9 + 3 + k];
                e->params[i].hi[k] = params[i * 9 + 6 + k];
            }
    uint32_t k0 = 0;
    mt_init_by_array(&e->rng, &k0, 1);
    /* MiniWorldEnv.__init__ ends with self.reset() (miniworld.py:523): the episode counters of the T-maze
     * family have seen one reset when the caller gets the env */
    if ((task == MWO_TMAZE && e->task_args[3] > 0) || (task == MWO_TMAZE_TWOBOX && e->task_args[0] == 0)) e->episode_count = 1;
    return e;
}

static void free_rooms(MwoEnv *e) {
    for (int i = 0; i < e->n_rooms; i++) { free(e->rooms[i].quads); free(e->rooms[i].segs); e->rooms[i].quads = NULL; e->rooms[i].segs = NULL; }
    free(e->wall_segs); e->wall_segs = NULL;
    e->n_rooms = 0; e->n_segs = 0; e->static_done = 0;
}

void mwo_destroy(MwoEnv *e) { if (!e) return; free_rooms(e); free(e); }

void mwo_seed_key(MwoEnv *e, const uint32_t *key, int n) { mt_init_by_array(&e->rng, key, n); }

static double norm3(const double *v) { /* np.linalg.norm(axis=1): sqrt(add.reduce(v*v)) */
    return sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
}

/* Room.__init__, miniworld.py:75-138 */
static Room *add_room_n(MwoEnv *e, const double outline2[4][2], int n_edges, double wall_height, int wall_tex, int floor_tex, int ceil_tex, int no_ceiling) {
    if (e->static_done) fail("cannot add rooms after static data is generated");
    if (e->n_rooms >= MWO_MAX_ROOMS) fail("too many rooms");
    Room *r = &e->rooms[e->n_rooms++];
    memset(r, 0, sizeof(*r));
    r->n_edges = n_edges;
    for (int i = 0; i < 4; i++) r->nbr[i] = -1;
    for (int i = 0; i < n_edges; i++) { r->outline[i][0] = outline2[i][0]; r->outline[i][1] = 0; r->outline[i][2] = outline2[i][1]; }
    r->min_x = r->max_x = r->outline[0][0]; r->min_z = r->max_z = r->outline[0][2];
    for (int i = 1; i < n_edges; i++) {
        if (r->outline[i][0] < r->min_x) r->min_x = r->outline[i][0];
        if (r->outline[i][0] > r->max_x) r->max_x = r->outline[i][0];
        if (r->outline[i][2] < r->min_z) r->min_z = r->outline[i][2];
        if (r->outline[i][2] > r->max_z) r->max_z = r->outline[i][2];
    }
    r->mid_x = (r->max_x + r->min_x) / 2; r->mid_z = (r->max_z + r->min_z) / 2;
    r->area = (r->max_x - r->min_x) * (r->max_z - r->min_z);
    for (int i = 0; i < n_edges; i++) {
        const double *p0 = r->outline[i], *p1 = r->outline[(i + 1) % n_edges];
        double d[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        double n = norm3(d);
        for (int k = 0; k < 3; k++) r->edge_dirs[i][k] = d[k] / n;
        /* -cross(edge_dir, Y_VEC) then normalise (miniworld.py:119-120) */
        const double *ed = r->edge_dirs[i];
        double c[3] = {ed[1] * 0 - ed[2] * 1, ed[2] * 0 - ed[0] * 0, ed[0] * 1 - ed[1] * 0};
        double en[3] = {-c[0], -c[1], -c[2]};
        double nn = norm3(en);
        for (int k = 0; k < 3; k++) r->edge_norms[i][k] = en[k] / nn;
    }
    r->wall_height = wall_height; r->no_ceiling = no_ceiling;
    r->wall_tex_name = wall_tex; r->floor_tex_name = floor_tex; r->ceil_tex_name = ceil_tex;
    return r;
}

static Room *add_room(MwoEnv *e, const double outline2[4][2], double wall_height, int wall_tex, int floor_tex, int ceil_tex, int no_ceiling) {
    return add_room_n(e, outline2, 4, wall_height, wall_tex, floor_tex, ceil_tex, no_ceiling);
}

/* add_rect_room, miniworld.py:718-743 */
static Room *add_rect_room(MwoEnv *e, double min_x, double max_x, double min_z, double max_z, int wall_tex) {
    double o[4][2] = {{max_x, max_z}, {max_x, min_z}, {min_x, min_z}, {min_x, max_z}};
    return add_room(e, o, 2.74, wall_tex, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES, 0);
}

static Room *add_rect_room_ex(MwoEnv *e, double min_x, double max_x, double min_z, double max_z, int wall_tex, int floor_tex, int ceil_tex, int no_ceiling) {
    double o[4][2] = {{max_x, max_z}, {max_x, min_z}, {min_x, min_z}, {min_x, max_z}};
    return add_room(e, o, 2.74, wall_tex, floor_tex, ceil_tex, no_ceiling);
}

/* Room.add_portal, miniworld.py:140-218.  mode 0: start/end given; 1: min_x/max_x; 2: min_z/max_z */
static void add_portal(Room *r, int edge, int mode, double a, double b, int has_max_y, double max_y_in, double *start_out, double *end_out) {
    double max_y = has_max_y ? max_y_in : r->wall_height, min_y = 0;
    const double *p0 = r->outline[edge], *p1 = r->outline[(edge + 1) % r->n_edges];
    double diff[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    /* 1-D np.linalg.norm = sqrt(dot(x,x)); exact for the axis-aligned edges of every configured task */
    double e_len = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
    double dx = diff[0] / e_len, dz = diff[2] / e_len;
    double start, end;
    if (mode == 1) {
        double m0 = (a - p0[0]) / dx, m1 = (b - p0[0]) / dx;
        if (m1 < m0) { double t = m0; m0 = m1; m1 = t; }
        start = m0; end = m1;
    } else if (mode == 2) {
        double m0 = (a - p0[2]) / dz, m1 = (b - p0[2]) / dz;
        if (m1 < m0) { double t = m0; m0 = m1; m1 = t; }
        start = m0; end = m1;
    } else { start = a; end = b; }
    if (!(end > start) || !(start >= 0) || !(end <= e_len)) fail("portal outside of wall extents");
    int n = r->n_portals[edge];
    if (n >= MWO_MAX_PORTALS) fail("too many portals on one edge");
    Portal np_ = {start, end, min_y, max_y};
    /* list.append + stable sort by start_pos */
    int k = n;
    r->portals[edge][n] = np_;
    while (k > 0 && r->portals[edge][k - 1].start > r->portals[edge][k].start) {
        Portal t = r->portals[edge][k - 1]; r->portals[edge][k - 1] = r->portals[edge][k]; r->portals[edge][k] = t; k--;
    }
    r->n_portals[edge]++;
    if (start_out) { *start_out = start; *end_out = end; }
}

/* connect_rooms, miniworld.py:757-843 */
static void connect_rooms(MwoEnv *e, int ia, int ib, int mode, double lo, double hi, int has_max_y, double max_y) {
    Room *A = &e->rooms[ia], *B = &e->rooms[ib];
    int idx_a = -1, idx_b = -1;
    for (int i = 0; i < A->n_edges && idx_a < 0; i++)
        for (int j = 0; j < B->n_edges; j++) {
            const double *na = A->edge_norms[i], *nb = B->edge_norms[j];
            double dotn = na[0] * nb[0] + na[1] * nb[1] + na[2] * nb[2];
            if (dotn > -0.9) continue;
            double dir[3] = {B->outline[j][0] - A->outline[i][0], B->outline[j][1] - A->outline[i][1], B->outline[j][2] - A->outline[i][2]};
            double dd = na[0] * dir[0] + na[1] * dir[1] + na[2] * dir[2];
            if (dd > 0.05) continue;
            idx_a = i; idx_b = j; break;
        }
    if (idx_a < 0) fail("matching edges not found in connect_rooms");
    double sa, ea, sb, eb;
    add_portal(A, idx_a, mode, lo, hi, has_max_y, max_y, &sa, &ea);
    add_portal(B, idx_b, mode, lo, hi, has_max_y, max_y, &sb, &eb);
    double a[3], b[3], c[3], d[3];
    for (int k = 0; k < 3; k++) {
        a[k] = A->outline[idx_a][k] + A->edge_dirs[idx_a][k] * sa;
        b[k] = A->outline[idx_a][k] + A->edge_dirs[idx_a][k] * ea;
        c[k] = B->outline[idx_b][k] + B->edge_dirs[idx_b][k] * sb;
        d[k] = B->outline[idx_b][k] + B->edge_dirs[idx_b][k] * eb;
    }
    double ad[3] = {a[0] - d[0], a[1] - d[1], a[2] - d[2]};
    if (sqrt(ad[0] * ad[0] + ad[1] * ad[1] + ad[2] * ad[2]) < 0.001) { A->nbr[idx_a] = ib; B->nbr[idx_b] = ia; return; }
    double ba[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, dc[3] = {d[0] - c[0], d[1] - c[1], d[2] - c[2]};
    double len_a = sqrt(ba[0] * ba[0] + ba[1] * ba[1] + ba[2] * ba[2]);
    double len_b = sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
    double o[4][2] = {{c[0], c[2]}, {b[0], b[2]}, {a[0], a[2]}, {d[0], d[2]}};
    double my = has_max_y ? max_y : A->wall_height;
    int wall_tex = A->wall_tex_name, floor_tex = A->floor_tex_name, ceil_tex = A->ceil_tex_name, noc = A->no_ceiling;
    Room *R = add_room(e, o, my, wall_tex, floor_tex, ceil_tex, noc);
    add_portal(R, 1, 0, 0, len_a, 0, 0, NULL, NULL);
    add_portal(R, 3, 0, 0, len_b, 0, 0, NULL, NULL);
    A = &e->rooms[ia]; B = &e->rooms[ib];
    A->nbr[idx_a] = e->n_rooms - 1; B->nbr[idx_b] = e->n_rooms - 1; R->nbr[1] = ia; R->nbr[3] = ib;
}

/* Texture.get, opengl.py:40-69: variant 1 unless rng, then rng.int(0, n) */
static int tex_get(MwoEnv *e, int family, int use_rng) {
    int idx = use_rng ? (int)rs_randint(&e->rng, 0, TEXF_COUNT[family]) : 0;
    return TEXF_FIRST[family] + idx;
}

static int tex_width(int id) { /* fallback when the pixels were not loaded: the sizes of the reference's PNGs */
    if (g_tex[id].w > 0) return g_tex[id].w;
    return (id == 3 || id == 5) ? 768 : (id == 6 || id == 9 || id == 13) ? 1024 : id == 10 ? 256 : 512;
}
static int tex_height(int id) { return g_tex[id].h > 0 ? g_tex[id].h : tex_width(id); }

typedef struct { Room *r; Quad *q; int nq, capq; double (*s)[2][3]; int ns, caps; } GenCtx;

/* gen_seg_poly, miniworld.py:267-309 */
static void gen_seg_poly(GenCtx *g, const double *edge_p0, const double *side_vec, double seg_start, double seg_end, double min_y, double max_y) {
    if (seg_end == seg_start) return;
    if (min_y == max_y) return;
    double s_p0[3], s_p1[3];
    for (int k = 0; k < 3; k++) { s_p0[k] = edge_p0[k] + seg_start * side_vec[k]; s_p1[k] = edge_p0[k] + seg_end * side_vec[k]; }
    if (min_y == 0) {
        if (g->ns == g->caps) { g->caps = g->caps ? g->caps * 2 : 8; g->s = realloc(g->s, sizeof(double[2][3]) * g->caps); }
        memcpy(g->s[g->ns][0], s_p1, sizeof(s_p1)); memcpy(g->s[g->ns][1], s_p0, sizeof(s_p0)); g->ns++;
    }
    if (g->nq == g->capq) { g->capq = g->capq ? g->capq * 2 : 8; g->q = realloc(g->q, sizeof(Quad) * g->capq); }
    Quad *q = &g->q[g->nq++];
    const double Y[3] = {0, 1, 0};
    for (int k = 0; k < 3; k++) {
        q->verts[0][k] = s_p0[k] + min_y * Y[k]; q->verts[1][k] = s_p0[k] + max_y * Y[k];
        q->verts[2][k] = s_p1[k] + max_y * Y[k]; q->verts[3][k] = s_p1[k] + min_y * Y[k];
    }
    double d[3] = {s_p1[0] - s_p0[0], s_p1[1] - s_p0[1], s_p1[2] - s_p0[2]};
    double c[3] = {d[1] * 0 - d[2] * 1, d[2] * 0 - d[0] * 0, d[0] * 1 - d[1] * 0};
    double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int k = 0; k < 3; k++) q->norm[k] = -c[k] / n;
    /* gen_texcs_wall, miniworld.py:19-46 (float32 result) */
    double xc = 512.0 / tex_width(g->r->wall_tex), yc = 512.0 / tex_height(g->r->wall_tex);
    double width = seg_end - seg_start, height = max_y - min_y;
    double min_u = seg_start * xc, max_u = (seg_start + width) * xc, min_v = min_y * yc, max_v = (min_y + height) * yc;
    q->texcs[0][0] = (float)min_u; q->texcs[0][1] = (float)min_v; q->texcs[1][0] = (float)min_u; q->texcs[1][1] = (float)max_v;
    q->texcs[2][0] = (float)max_u; q->texcs[2][1] = (float)max_v; q->texcs[3][0] = (float)max_u; q->texcs[3][1] = (float)min_v;
}

/* Room._gen_static_data, miniworld.py:234-388 */
static void room_gen_static(MwoEnv *e, Room *r, int use_rng) {
    r->wall_tex = tex_get(e, r->wall_tex_name, use_rng);
    r->floor_tex = tex_get(e, r->floor_tex_name, use_rng);
    r->ceil_tex = tex_get(e, r->ceil_tex_name, use_rng);
    /* gen_texcs_floor, miniworld.py:48-68 */
    for (int i = 0; i < r->n_edges; i++) {
        r->floor_texcs[i][0] = r->outline[i][0] * (512.0 / tex_width(r->floor_tex));
        r->floor_texcs[i][1] = r->outline[i][2] * (512.0 / tex_height(r->floor_tex));
        const double *cv = r->outline[r->n_edges - 1 - i]; /* np.flip(outline, axis=0) */
        r->ceil_texcs[i][0] = (cv[0] + r->wall_height * 0) * (512.0 / tex_width(r->ceil_tex));
        r->ceil_texcs[i][1] = (cv[2] + r->wall_height * 0) * (512.0 / tex_height(r->ceil_tex));
    }
    GenCtx g = {r, NULL, 0, 0, NULL, 0, 0};
    for (int w = 0; w < r->n_edges; w++) {
        const double *p0 = r->outline[w], *p1 = r->outline[(w + 1) % r->n_edges];
        double d[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        double wall_width = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        double side[3] = {d[0] / wall_width, d[1] / wall_width, d[2] / wall_width};
        int np_ = r->n_portals[w];
        double seg_end = np_ > 0 ? r->portals[w][0].start : wall_width;
        gen_seg_poly(&g, p0, side, 0, seg_end, 0, r->wall_height);
        for (int k = 0; k < np_; k++) {
            Portal *p = &r->portals[w][k];
            gen_seg_poly(&g, p0, side, p->start, p->end, 0, p->min_y);
            gen_seg_poly(&g, p0, side, p->start, p->end, p->max_y, r->wall_height);
            double next_start = (k < np_ - 1) ? r->portals[w][k + 1].start : wall_width;
            gen_seg_poly(&g, p0, side, p->end, next_start, 0, r->wall_height);
        }
    }
    r->quads = g.q; r->n_quads = g.nq; r->segs = g.s; r->n_segs = g.ns;
}

/* MiniWorldEnv._gen_static_data, miniworld.py:981-998 */
static void gen_static_data(MwoEnv *e) {
    int total = 0;
    for (int i = 0; i < e->n_rooms; i++) { room_gen_static(e, &e->rooms[i], e->domain_rand); total += e->rooms[i].n_segs; }
    e->wall_segs = malloc(sizeof(double[2][3]) * (total > 0 ? total : 1));
    int k = 0;
    for (int i = 0; i < e->n_rooms; i++)
        for (int s = 0; s < e->rooms[i].n_segs; s++) memcpy(e->wall_segs[k++], e->rooms[i].segs[s], sizeof(double[2][3]));
    e->n_segs = total;
    double areas[MWO_MAX_ROOMS];
    for (int i = 0; i < e->n_rooms; i++) areas[i] = e->rooms[i].area;
    double sum = np_pairwise_sum(areas, e->n_rooms);
    for (int i = 0; i < e->n_rooms; i++) e->room_probs[i] = areas[i] / sum;
    /* RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1] */
    double acc = 0;
    for (int i = 0; i < e->n_rooms; i++) { acc = (i == 0) ? e->room_probs[0] : acc + e->room_probs[i]; e->room_cdf[i] = acc; }
    double last = e->room_cdf[e->n_rooms - 1];
    for (int i = 0; i < e->n_rooms; i++) e->room_cdf[i] /= last;
    e->static_done = 1;
}

/* math.py:25-57 */
int mwo_intersect_circle_segs(const double *point, double radius, const double *segs, int n) {
    double px = point[0], pz = point[2];
    for (int i = 0; i < n; i++) {
        const double *a = segs + i * 6, *b = a + 3;
        double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        double ap[3] = {px - a[0], 0 - a[1], pz - a[2]};
        double dotAPAB = (ap[0] * ab[0] + ap[1] * ab[1]) + ap[2] * ab[2];
        double dotABAB = (ab[0] * ab[0] + ab[1] * ab[1]) + ab[2] * ab[2];
        double proj = dotAPAB / dotABAB;
        if (proj < 0) proj = 0; else if (proj > 1) proj = 1; /* np.clip; nan propagates */
        double c[3] = {a[0] + proj * ab[0], a[1] + proj * ab[1], a[2] + proj * ab[2]};
        double dv[3] = {c[0] - px, c[1] - 0, c[2] - pz};
        double dist = sqrt((dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]);
        if (dist < radius) return 1;
    }
    return 0;
}

/* MiniWorldEnv.intersect, miniworld.py:933-959. returns 1 wall, 2+k entity k, 0 none */
/* a + b as the reference's interpreter evaluates it when either operand may be a numpy float32 SCALAR and the other a Python
 * float (NumPy >= 2, NEP 50): the Python float is cast to float32 and the sum is a float32; two Python floats add in float64 */
static double tagged_add(double a, int a_f32, double b, int b_f32, int *res_f32) {
    if (a_f32 || b_f32) { if (res_f32) *res_f32 = 1; return (double)((float)a + (float)b); }
    if (res_f32) *res_f32 = 0;
    return a + b;
}

/* MiniWorldEnv.intersect(ent, pos, radius), miniworld.py:933-959; radius_f32: `radius` is a numpy float32 scalar.
 * Returns 0 none, 1 wall, 2 + slot for an entity, 2 + n_boxes for the agent */
static int intersect_t(MwoEnv *e, const Ent *self, const double *pos, double radius, int radius_f32) {
    double p[3] = {pos[0], 0, pos[2]};
    if (mwo_intersect_circle_segs(p, radius, &e->wall_segs[0][0][0], e->n_segs)) return 1;
    for (int k = 0; k < e->n_order; k++) { /* self.entities in list order */
        const int slot = e->order[k];
        const Ent *o = slot == AGENT_SLOT ? &e->agent : &e->boxes[slot];
        if (o == self) continue;
        double d3[3] = {o->pos[0] - p[0], 0 - p[1], o->pos[2] - p[2]}; /* ent2's y is flattened too (miniworld.py:951-952) */
        double d = sqrt(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
        if (d < tagged_add(radius, radius_f32, o->radius, o->rad_f32, NULL)) return 2 + (slot == AGENT_SLOT ? e->n_boxes : slot);
    }
    return 0;
}
static int intersect(MwoEnv *e, const Ent *self, const double *pos, double radius) { return intersect_t(e, self, pos, radius, 0); }

/* ent: index in the entity list (boxes, then the agent = n_boxes); legacy callers pass 0 = first box, 1 = agent, 2 = second box */
int mwo_intersect(MwoEnv *e, int ent, double x, double z, double radius) {
    double p[3] = {x, 0, z};
    const Ent *self = ent == 0 ? &e->boxes[0] : ent == 2 ? &e->boxes[1] : &e->agent;
    return intersect(e, self, p, radius);
}
int mwo_intersect_ent(MwoEnv *e, int ent_index, double x, double z, double radius) {
    double p[3] = {x, 0, z};
    const Ent *self = ent_index < 0 ? NULL : ent_index < e->n_boxes ? &e->boxes[ent_index] : &e->agent;
    return intersect(e, self, p, radius);
}

/* Room.point_inside, miniworld.py:220-232 */
static int point_inside(const Room *r, const double *p) {
    for (int i = 0; i < r->n_edges; i++) {
        double ap[3] = {p[0] - r->outline[i][0], p[1] - r->outline[i][1], p[2] - r->outline[i][2]};
        const double *n = r->edge_norms[i];
        double dot = (n[0] * ap[0] + n[1] * ap[1]) + n[2] * ap[2];
        if (!(dot > 0)) return 0;
    }
    return 1;
}

#define NOVAL (-1e300)
/* place_entity, miniworld.py:845-907 (pos=None path) */
static void place_entity_in(MwoEnv *e, Ent *ent, int room, int has_dir, double dir, double min_x, double max_x, double min_z, double max_z) {
    if (e->n_rooms <= 0) fail("create rooms before calling place_entity");
    if (!e->static_done) gen_static_data(e);
    for (;;) {
        Room *r = room >= 0 ? &e->rooms[room] : &e->rooms[rs_choice_cdf(&e->rng, e->room_cdf, e->n_rooms)];
        double lx = min_x == NOVAL ? r->min_x : min_x, hx = max_x == NOVAL ? r->max_x : max_x;
        double lz = min_z == NOVAL ? r->min_z : min_z, hz = max_z == NOVAL ? r->max_z : max_z;
        double pos[3];
        pos[0] = rs_uniform(&e->rng, lx + ent->radius, hx - ent->radius);
        pos[1] = rs_uniform(&e->rng, 0, 0);
        pos[2] = rs_uniform(&e->rng, lz + ent->radius, hz - ent->radius);
        if (!point_inside(r, pos)) continue;
        if (intersect_t(e, ent, pos, ent->radius, ent->rad_f32)) continue;
        double d = has_dir ? dir : rs_uniform(&e->rng, -M_PI, M_PI);
        memcpy(ent->pos, pos, sizeof(pos));
        ent->dir = d;
        break;
    }
    e->order[e->n_order++] = ent == &e->agent ? AGENT_SLOT : (int)(ent - e->boxes);   /* self.entities.append(ent) */
}

/* place_entity with pos given (miniworld.py:869-873): no tests, the heading drawn unless given */
static void place_entity_at(MwoEnv *e, Ent *ent, double x, double y, double z, int has_dir, double dir) {
    if (!e->static_done) gen_static_data(e);
    ent->dir = has_dir ? dir : rs_uniform(&e->rng, -M_PI, M_PI);
    ent->pos[0] = x; ent->pos[1] = y; ent->pos[2] = z;
    e->order[e->n_order++] = ent == &e->agent ? AGENT_SLOT : (int)(ent - e->boxes);
}

static void place_entity(MwoEnv *e, Ent *ent, int has_dir, double dir, double min_x, double max_x, double min_z, double max_z) {
    place_entity_in(e, ent, -1, has_dir, dir, min_x, max_x, min_z, max_z);
}

/* entity.py:362-379 Box(color, size=s): radius = sqrt(sx^2 + sz^2) / 2, height = sy */
static void size_box(MwoEnv *e, int b, double s) {
    Ent *ent = &e->boxes[b];
    e->box_s[b] = s;
    ent->radius = sqrt(s * s + s * s) / 2;
    ent->height = s;
    ent->kind = MWO_ENT_BOX; ent->is_static = 0; ent->alive = 1; ent->rad_f32 = 0;
}
/* MeshEnt(mesh_name, height, static) (entity.py:100-128): scale / radius as registered with mwo_set_mesh_dims */
static void make_mesh_ent(MwoEnv *e, int b, int geom, double height, int is_static, int color) {
    Ent *ent = &e->boxes[b];
    const MeshDims *md = NULL;
    for (int i = 0; i < g_n_mesh_dims; i++) if (g_mesh_dims[i].geom == geom && g_mesh_dims[i].height == height) md = &g_mesh_dims[i];
    if (!md) fail("mesh dimensions not registered (mwo_set_mesh_dims)");
    memset(ent, 0, sizeof(*ent));
    ent->kind = MWO_ENT_MESH; ent->geom = geom; ent->is_static = is_static; ent->alive = 1;
    ent->scale = md->scale; ent->radius = md->radius; ent->rad_f32 = md->is_f32; ent->height = height;
    e->box_s[b] = height; e->box_base[b] = color;   /* colour index into COLORS: the mesh's Kd (ball_<c>.mtl / key_<c>.mtl); -1 = white */
}
static void new_box(MwoEnv *e) { for (int b = 0; b < MWO_MAX_BOXES; b++) size_box(e, b, 0.8); }

/* one row of np.dot(A (N x 3), M (3 x 3)) as numpy's BLAS computes it: see the note at its definition */
static void mat_row_dot(const double *v, const double *m, double *out);

/* envs/maze.py:34-104 */
typedef struct { int i, j; int order[4][2]; int next; } Frame;
static void gen_maze(MwoEnv *e) {
    int num_rows = (int)e->task_args[0], num_cols = (int)e->task_args[1];
    double room_size = e->task_args[2], gap = 0.25;
    for (int j = 0; j < num_rows; j++)
        for (int i = 0; i < num_cols; i++) {
            double min_x = i * (room_size + gap), max_x = min_x + room_size;
            double min_z = j * (room_size + gap), max_z = min_z + room_size;
            add_rect_room(e, min_x, max_x, min_z, max_z, TEXF_BRICK_WALL);
        }
    char *visited = calloc(num_rows * num_cols, 1);
    Frame *stack = malloc(sizeof(Frame) * (num_rows * num_cols + 1));
    int sp = 0;
    static const int NB[4][2] = {{0, 1}, {0, -1}, {-1, 0}, {1, 0}}; /* (dj, di) */
#define PUSH(I, J) do { Frame *f = &stack[sp++]; f->i = (I); f->j = (J); f->next = 0; visited[(J) * num_cols + (I)] = 1; \
        int lst[4] = {0, 1, 2, 3}, n = 4; \
        for (int q = 0; q < 4; q++) { /* RandGen.subset: choice + list.remove, random.py:50-65 */ \
            int idx = (int)rs_randint(&e->rng, 0, n); \
            f->order[q][0] = NB[lst[idx]][0]; f->order[q][1] = NB[lst[idx]][1]; \
            for (int w = idx; w < n - 1; w++) { lst[w] = lst[w + 1]; } \
            n--; } } while (0)
    PUSH(0, 0);
    while (sp > 0) {
        Frame *f = &stack[sp - 1];
        if (f->next >= 4) { sp--; continue; }
        int dj = f->order[f->next][0], di = f->order[f->next][1];
        f->next++;
        int ni = f->i + di, nj = f->j + dj;
        if (nj < 0 || nj >= num_rows || ni < 0 || ni >= num_cols) continue;
        if (visited[nj * num_cols + ni]) continue;
        int ra = f->j * num_cols + f->i, rb = nj * num_cols + ni;
        Room *room = &e->rooms[ra];
        if (di == 0) connect_rooms(e, ra, rb, 1, room->min_x, room->max_x, 0, 0);
        else if (dj == 0) connect_rooms(e, ra, rb, 2, room->min_z, room->max_z, 0, 0);
        PUSH(ni, nj);
    }
#undef PUSH
    free(visited); free(stack);
    new_box(e);
    place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
    place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
}

static void gen_world(MwoEnv *e) {
    switch (e->task) {
    case MWO_HALLWAY: { /* envs/hallway.py:25-42 */
        double length = e->task_args[0];
        Room *room = add_rect_room(e, -1, -1 + length, -2, 2, TEXF_CONCRETE);
        double rmax = room->max_x;
        new_box(e);
        place_entity(e, &e->boxes[0], 0, 0, rmax - 2, NOVAL, NOVAL, NOVAL);
        double dir = rs_uniform(&e->rng, -M_PI / 4, M_PI / 4);
        place_entity(e, &e->agent, 1, dir, NOVAL, rmax - 2, NOVAL, NOVAL);
        break;
    }
    case MWO_ONEROOM: { /* envs/oneroom.py:26-35 */
        double size = e->task_args[0];
        add_rect_room(e, 0, size, 0, size, TEXF_CONCRETE);
        new_box(e);
        place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_FOURROOMS: { /* envs/fourrooms.py:22-52 */
        add_rect_room(e, -7, -1, 1, 7, TEXF_CONCRETE);
        add_rect_room(e, 1, 7, 1, 7, TEXF_CONCRETE);
        add_rect_room(e, 1, 7, -7, -1, TEXF_CONCRETE);
        add_rect_room(e, -7, -1, -7, -1, TEXF_CONCRETE);
        connect_rooms(e, 0, 1, 2, 3, 5, 1, 2.2);
        connect_rooms(e, 1, 2, 1, 3, 5, 1, 2.2);
        connect_rooms(e, 2, 3, 2, -5, -3, 1, 2.2);
        connect_rooms(e, 3, 0, 1, -5, -3, 1, 2.2);
        new_box(e);
        place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_MAZE: gen_maze(e); break;
    case MWO_TMAZE:          /* envs/tmaze.py:27-61 (TMaze, TMazeLeft/Right, TMazeDynamic) */
    case MWO_TMAZE_TWOBOX: { /* envs/tmaze.py:151-194 and the *Features* copies */
        add_rect_room(e, -1, 8, -2, 2, TEXF_CONCRETE);
        Room *room2 = add_rect_room(e, 8, 12, -8, 8, TEXF_CONCRETE);
        double r2min = room2->min_z, r2max = room2->max_z;
        connect_rooms(e, 0, 1, 2, -2, 2, 0, 0);
        new_box(e);
        if (e->task == MWO_TMAZE_TWOBOX) {
            place_entity(e, &e->boxes[0], 0, 0, 10, 10, -6, -6);
            place_entity(e, &e->boxes[1], 0, 0, 10, 10, 6, 6);
        } else if (e->task_args[0] != 0) { /* goal_pos given */
            double gx = e->task_args[1], gz = e->task_args[2];
            if (e->task_args[3] > 0) { gx = 10; gz = e->goal_idx ? 6 : -6; } /* TMazeDynamic.goals, tmaze.py:88 */
            place_entity(e, &e->boxes[0], 0, 0, gx, gx, gz, gz);
        } else if (rs_randint(&e->rng, 0, 2) == 0) /* RandGen.bool, random.py:26-31 */
            place_entity_in(e, &e->boxes[0], 1, 0, 0, NOVAL, NOVAL, NOVAL, r2min + 2);
        else
            place_entity_in(e, &e->boxes[0], 1, 0, 0, NOVAL, NOVAL, r2max - 2, NOVAL);
        double dir = rs_uniform(&e->rng, -M_PI / 4, M_PI / 4);
        place_entity_in(e, &e->agent, 0, 1, dir, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_SIM2REAL_GOTO:   /* envs/simtorealgoto.py:40-82 */
    case MWO_SIM2REAL_PUSH: { /* envs/simtorealpush.py:39-107 */
        const int push = e->task == MWO_SIM2REAL_PUSH;
        double size = push ? rs_uniform(&e->rng, 1.6, 1.7) : rs_uniform(&e->rng, 1, 2);
        double wall_height = push ? rs_uniform(&e->rng, 0.42, 0.50) : rs_uniform(&e->rng, 0.20, 0.50);
        double s1 = push ? rs_uniform(&e->rng, 0.075, 0.090) : rs_uniform(&e->rng, 0.07, 0.12);
        double s2 = push ? rs_uniform(&e->rng, 0.075, 0.090) : 0.8;
        e->agent.radius = 0.11;
        static const int FLOORS[3] = {TEXF_CARDBOARD, TEXF_WOOD, TEXF_WOOD_PLANKS};
        static const int WALLS_GOTO[5] = {TEXF_DRYWALL, TEXF_STUCCO, TEXF_CARDBOARD, TEXF_CONCRETE_TILES, TEXF_CEILING_TILES};
        static const int WALLS_PUSH[4] = {TEXF_DRYWALL, TEXF_STUCCO, TEXF_CONCRETE_TILES, TEXF_CEILING_TILES};
        int floor_tex = FLOORS[rs_randint(&e->rng, 0, 3)]; /* RandGen.choice -> np_random.choice(len) -> randint */
        int wall_tex = push ? WALLS_PUSH[rs_randint(&e->rng, 0, 4)] : WALLS_GOTO[rs_randint(&e->rng, 0, 5)];
        double o[4][2] = {{size, size}, {size, 0}, {0, 0}, {0, size}};
        add_room(e, o, wall_height, wall_tex, floor_tex, TEXF_CONCRETE_TILES, 1);
        size_box(e, 0, s1); size_box(e, 1, s2);
        if (push) {
            e->goal_dist = 1.5 * (s1 + s2);
            double min_pos = 2 * e->params[MWO_P_BOT_RADIUS].hi[0], max_pos = size - 2 * e->params[MWO_P_BOT_RADIUS].hi[0];
            for (;;) { /* boxes can't start too close to each other */
                place_entity(e, &e->boxes[0], 0, 0, min_pos, max_pos, min_pos, max_pos);
                place_entity(e, &e->boxes[1], 0, 0, min_pos, max_pos, min_pos, max_pos);
                double dx = e->boxes[0].pos[0] - e->boxes[1].pos[0], dy = e->boxes[0].pos[1] - e->boxes[1].pos[1], dz = e->boxes[0].pos[2] - e->boxes[1].pos[2];
                if (sqrt((dx * dx + dy * dy) + dz * dz) > e->goal_dist) break;
                e->n_order = 0; /* entities.remove(box1), entities.remove(box2) */
            }
        } else {
            place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        }
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_PUTNEXT: { /* envs/putnext.py:22-43 */
        double size = e->task_args[0];
        add_rect_room(e, 0, size, 0, size, TEXF_CONCRETE);
        for (int b = 0; b < 6; b++) { /* Box(color=color, size=self.rand.float(0.6, 0.85)): the size is drawn before the placement */
            size_box(e, b, rs_uniform(&e->rng, 0.6, 0.85));
            place_entity(e, &e->boxes[b], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        }
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_YMAZE: { /* envs/ymaze.py:28-83 */
        const double main_o[4][2] = {{-9.15, -2}, {-9.15, 2}, {-1.15, 2}, {-1.15, -2}};
        add_room(e, main_o, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES, 0);
        const double hub_o[4][2] = {{-1.15, -2}, {-1.15, 2}, {2.31, 0}, {0, 0}};
        add_room_n(e, hub_o, 3, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES, 0);
        const double Yv[3] = {0, 1, 0};
        for (int arm = 0; arm < 2; arm++) { /* np.dot(main_outline, gen_rot_matrix(Y, -+120 deg)) */
            double m[9], o[4][2];
            mwo_gen_rot_matrix(Yv, (arm == 0 ? -120 : 120) * (M_PI / 180), m);
            for (int i = 0; i < 4; i++) {
                const double v[3] = {main_o[i][0], 0, main_o[i][1]};
                double out[3];
                mat_row_dot(v, m, out);
                o[i][0] = out[0]; o[i][1] = out[2];
            }
            add_room(e, o, 2.74, TEXF_CONCRETE, TEXF_FLOOR_TILES_BW, TEXF_CONCRETE_TILES, 0);
        }
        connect_rooms(e, 0, 1, 2, -2, 2, 0, 0);
        connect_rooms(e, 2, 1, 2, -1.995, 0, 0, 0);
        connect_rooms(e, 3, 1, 2, 0, 1.995, 0, 0);
        new_box(e);
        if (e->task_args[0] != 0) { /* goal_pos given (YMazeLeft / YMazeRight, ymaze.py:97-103) */
            double gx = e->task_args[1], gz = e->task_args[2];
            place_entity(e, &e->boxes[0], 0, 0, gx, gx, gz, gz);
        } else if (rs_randint(&e->rng, 0, 2) == 0) /* RandGen.bool */
            place_entity_in(e, &e->boxes[0], 2, 0, 0, NOVAL, NOVAL, NOVAL, e->rooms[2].min_z + 2.5);
        else
            place_entity_in(e, &e->boxes[0], 3, 0, 0, NOVAL, NOVAL, e->rooms[3].max_z - 2.5, NOVAL);
        double dir = rs_uniform(&e->rng, -M_PI / 4, M_PI / 4);
        place_entity_in(e, &e->agent, 0, 1, dir, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_PICKUPOBJS: { /* envs/pickupobjs.py:28-54 */
        double size = e->task_args[0];
        add_rect_room_ex(e, 0, size, 0, size, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES, 1);
        for (int i = 0; i < e->n_boxes; i++) {
            int type = (int)rs_randint(&e->rng, 0, 3);    /* self.rand.choice([Ball, Box, Key]) */
            int color = (int)rs_randint(&e->rng, 0, 6);   /* self.rand.color(): choice(COLOR_NAMES) */
            if (type == 1) { size_box(e, i, 0.9); e->box_base[i] = color; }
            else if (type == 0) make_mesh_ent(e, i, MWO_MESH_BALL, 0.9, 0, color);
            else make_mesh_ent(e, i, MWO_MESH_KEY, 0.35, 0, color);
            place_entity(e, &e->boxes[i], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        }
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        e->num_picked = 0;
        break;
    }
    case MWO_ROOMOBJS: { /* envs/roomobjs.py:24-47 */
        double size = e->task_args[0];
        add_rect_room_ex(e, 0, size, 0, size, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES, 1);
        e->agent.radius = 1.5;   /* "Reduce chances that objects are too close to see" */
        int color = (int)rs_randint(&e->rng, 0, 6);   /* the colour is an argument of the constructor: drawn before the placement */
        size_box(e, 0, 0.9); e->box_base[0] = color;
        place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        color = (int)rs_randint(&e->rng, 0, 6);
        make_mesh_ent(e, 1, MWO_MESH_BALL, 0.9, 0, color);
        place_entity(e, &e->boxes[1], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        color = (int)rs_randint(&e->rng, 0, 6);
        make_mesh_ent(e, 2, MWO_MESH_KEY, 0.35, 0, color);
        place_entity(e, &e->boxes[2], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_COLLECTHEALTH: { /* envs/collecthealth.py:28-49 */
        double size = e->task_args[0];
        add_rect_room_ex(e, 0, size, 0, size, TEXF_CINDER_BLOCKS, TEXF_SLIME, TEXF_CONCRETE_TILES, 0);
        for (int i = 0; i < 18; i++) {
            make_mesh_ent(e, i, MWO_MESH_MEDKIT, 0.40, 0, -1);
            place_entity(e, &e->boxes[i], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        }
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        e->health = 100;
        break;
    }
    case MWO_THREEROOMS: { /* envs/threerooms.py:22-69 */
        add_rect_room(e, -7, 7, 0.5, 7, TEXF_CONCRETE);
        add_rect_room(e, -7, -1, -7, -0.5, TEXF_CONCRETE);
        add_rect_room(e, 1, 7, -7, -0.5, TEXF_CONCRETE);
        connect_rooms(e, 0, 1, 1, -5.25, -2.75, 0, 0);
        connect_rooms(e, 0, 2, 1, 2.75, 5.25, 0, 0);
        size_box(e, 0, 0.8); e->box_base[0] = 4;   /* red */
        place_entity(e, &e->boxes[0], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        size_box(e, 1, 0.6); e->box_base[1] = 1;   /* green */
        place_entity(e, &e->boxes[1], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        {   /* self.entities.append(ImageFrame(pos=[0, 1.35, 7], dir=pi/2, width=1.8, tex_name='logo_mila')) */
            Ent *f = &e->boxes[2];
            memset(f, 0, sizeof(*f));
            f->kind = MWO_ENT_IMAGE; f->is_static = 1; f->alive = 1; f->radius = 0; f->rad_f32 = 0;
            f->pos[0] = 0; f->pos[1] = 1.35; f->pos[2] = 7; f->dir = M_PI / 2;
            f->tex[0] = TEXF_FIRST[TEXF_LOGO_MILA];
            f->frame_w = 1.8; f->frame_d = 0.05;
            f->frame_h = ((double)tex_height(f->tex[0]) / tex_width(f->tex[0])) * 1.8;   /* (float(tex.height) / tex.width) * width */
            f->height = f->frame_h;
            e->order[e->n_order++] = 2;
        }
        make_mesh_ent(e, 3, MWO_MESH_DUCKIE, 0.25, 0, -1);
        place_entity(e, &e->boxes[3], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        make_mesh_ent(e, 4, MWO_MESH_KEY, 0.35, 0, 0);    /* Key(color='blue') */
        place_entity(e, &e->boxes[4], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        make_mesh_ent(e, 5, MWO_MESH_BALL, 0.6, 0, 1);    /* Ball(color='green'), size 0.6 */
        place_entity(e, &e->boxes[5], 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        place_entity(e, &e->agent, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    case MWO_SIGN: { /* envs/sign.py:75-113 */
        const double size = e->task_args[0], gap = 0.25;
        add_rect_room(e, 0, size, 0, size * 0.65, TEXF_CONCRETE);
        add_rect_room(e, 0, size * 3 / 5, size * 0.65 + gap, size * 1.3, TEXF_CONCRETE);
        add_rect_room(e, size * 3 / 5, size, size * 0.65 + gap, size * 1.3, TEXF_CONCRETE);
        connect_rooms(e, 0, 1, 1, 0, size * 3 / 5, 0, 0);
        connect_rooms(e, 1, 2, 2, size * 0.65 + gap, size * 1.3, 0, 0);
        static const int box_col[3] = {0, 4, 1};   /* blue, red, green */
        static const double box_at[3][2] = {{1, 1}, {9, 1}, {9, 5}}, key_at[3][2] = {{5, 1}, {1, 5}, {1, 9}};
        for (int i = 0; i < 3; i++) { size_box(e, i, 0.8); e->box_base[i] = box_col[i]; place_entity_at(e, &e->boxes[i], box_at[i][0], 0, box_at[i][1], 0, 0); }
        for (int i = 0; i < 3; i++) { make_mesh_ent(e, 3 + i, MWO_MESH_KEY, 0.6, 0, box_col[i]); place_entity_at(e, &e->boxes[3 + i], key_at[i][0], 0, key_at[i][1], 0, 0); }   /* BigKey */
        {   /* TextFrame(pos=[size, 1.35, size + gap], dir=pi, str=text, height=1) appended to the list */
            static const char *TEXT[3] = {"BLUE", "RED", "GREEN"};
            const char *txt = TEXT[(int)e->task_args[1]];
            Ent *f = &e->boxes[6];
            memset(f, 0, sizeof(*f));
            f->kind = MWO_ENT_TEXT; f->is_static = 1; f->alive = 1;
            f->pos[0] = size; f->pos[1] = 1.35; f->pos[2] = size + gap; f->dir = M_PI;
            f->n_chars = (int)strlen(txt);
            f->frame_h = 1; f->frame_d = 0.05; f->frame_w = f->n_chars * 1.0; f->height = 1;
            for (int c = 0; c < f->n_chars; c++) f->tex[c] = TEX_CHAR0 + 9 * (int)(strchr(SIGN_CHARS, txt[c]) - SIGN_CHARS);   /* variant 1 until randomize() */
            e->order[e->n_order++] = 6;
        }
        place_entity(e, &e->agent, 0, 0, 4, 5, 4, 6);
        break;
    }
    case MWO_SIDEWALK: { /* envs/sidewalk.py:22-72 */
        add_rect_room_ex(e, -3, 0, 0, 12, TEXF_BRICK_WALL, TEXF_CONCRETE_TILES, TEXF_CONCRETE_TILES, 1);
        add_rect_room_ex(e, 0, 6, -80, 80, TEXF_CONCRETE, TEXF_ASPHALT, TEXF_CONCRETE_TILES, 1);
        connect_rooms(e, 0, 1, 2, 0, 12, 0, 0);
        make_mesh_ent(e, 0, MWO_MESH_BUILDING, 30, 1, -1);
        place_entity_at(e, &e->boxes[0], 30, 0, 30, 1, -M_PI);
        for (int i = 1; i < 6; i++) {   /* range(1, sidewalk.max_z // 2) */
            make_mesh_ent(e, i, MWO_MESH_CONE, 0.75, 1, -1);
            place_entity_at(e, &e->boxes[i], 1, 0, 2 * i, 0, 0);
        }
        size_box(e, 6, 0.8); e->box_base[6] = 4;
        place_entity_in(e, &e->boxes[6], 0, 0, 0, NOVAL, NOVAL, e->rooms[0].max_z - 2, e->rooms[0].max_z);
        place_entity_in(e, &e->agent, 0, 0, 0, NOVAL, NOVAL, 0, 1.5);
        break;
    }
    case MWO_WALLGAP: { /* envs/wallgap.py:21-52 */
        add_rect_room_ex(e, -7, 7, 0.5, 8, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES, 1);
        add_rect_room_ex(e, -7, 7, -8, -0.5, TEXF_BRICK_WALL, TEXF_ASPHALT, TEXF_CONCRETE_TILES, 1);
        connect_rooms(e, 0, 1, 1, -1.5, 1.5, 0, 0);
        size_box(e, 0, 0.8); e->box_base[0] = 4;
        place_entity_in(e, &e->boxes[0], 1, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        make_mesh_ent(e, 1, MWO_MESH_BUILDING, 30, 1, -1);
        place_entity_at(e, &e->boxes[1], 30, 0, 30, 1, -M_PI);
        place_entity_in(e, &e->agent, 0, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
        break;
    }
    default: fail("unknown task");
    }
}

/* MiniWorldEnv.reset, miniworld.py:532-592 */
void mwo_reset(MwoEnv *e) {
    /* reset() overrides of the T-maze family run before MiniWorldEnv.reset */
    if (e->task == MWO_TMAZE && e->task_args[3] > 0) { /* TMazeDynamic.reset, tmaze.py:98-105 */
        e->episode_count += 1;
        if (e->episode_count % (long long)e->task_args[3] == 0) e->goal_idx = (e->goal_idx + 1) % 2;
    } else if (e->task == MWO_TMAZE_TWOBOX) {
        if (e->task_args[0] == 0) { /* TMazeTwoBoxDynamic.reset, tmaze.py:210-217 */
            e->episode_count += 1;
            if (e->episode_count % (long long)e->task_args[3] == 0) e->goal_idx = (e->goal_idx + 1) % 2;
        } else if ((double)e->task_step_count > e->task_args[3]) /* *Features*.reset, tmaze.py:322-330: the counter */
            e->goal_idx = (e->goal_idx + 1) % 2;                   /* is never cleared (the reference assigns a misspelt name) */
    }
    e->feature[0] = e->feature[1] = 0;
    e->step_count = 0;
    free_rooms(e);
    e->n_order = 0;
    memset(&e->agent, 0, sizeof(Ent));
    e->agent.radius = 0.4; e->agent.height = 1.6; /* entity.py:436-455 */
    e->carrying = -1; /* a fresh Agent() carries nothing (entity.py:446) */
    for (int b = 0; b < MWO_MAX_BOXES; b++) { memset(&e->boxes[b], 0, sizeof(Ent)); e->box_base[b] = 4; /* red */ e->box_s[b] = 0; }
    if (e->task == MWO_TMAZE_TWOBOX) e->box_base[1] = 0;          /* blue, tmaze.py:168 */
    else if (e->task == MWO_SIM2REAL_PUSH) e->box_base[1] = 5;    /* yellow, simtorealpush.py:93 */
    else if (e->task == MWO_PUTNEXT) for (int b = 0; b < 6; b++) e->box_base[b] = b;   /* for color in COLOR_NAMES, putnext.py:31 */
    gen_world(e);
    int dr = e->domain_rand;
    sample_param(e, dr, MWO_P_SKY_COLOR, e->sky_color);
    sample_param(e, dr, MWO_P_LIGHT_POS, e->light_pos);
    sample_param(e, dr, MWO_P_LIGHT_COLOR, e->light_color);
    sample_param(e, dr, MWO_P_LIGHT_AMBIENT, e->light_ambient);
    e->max_forward_step = e->params[MWO_P_FORWARD_STEP].hi[0];
    /* Box.randomize, entity.py:381-383: COLORS[color] + bias, clipped - entities are randomized in list order */
    static const double COLORS[6][3] = {{0.0, 0.0, 1.0}, {0.0, 1.0, 0.0}, {0.39, 0.39, 0.39}, {0.44, 0.15, 0.76},
                                        {1.0, 0.0, 0.0}, {1.00, 1.00, 0.00}}; /* COLOR_NAMES order: blue green grey purple red yellow */
    for (int k = 0; k < e->n_order; k++) {   /* for ent in self.entities: ent.randomize(params, rand) (miniworld.py:572-573) */
        const int b = e->order[k];
        if (b == AGENT_SLOT) continue;   /* the agent is the last entry in every task: its draws follow below */
        Ent *ent = &e->boxes[b];
        if (ent->kind == MWO_ENT_BOX) {
            double bias[3];
            sample_param(e, dr, MWO_P_OBJ_COLOR_BIAS, bias);
            const double *c = COLORS[e->box_base[b]];
            for (int q = 0; q < 3; q++) { double v = c[q] + bias[q]; e->box_colors[b][q] = v < 0 ? 0 : (v > 1 ? 1 : v); }
        } else if (ent->kind == MWO_ENT_MESH) {   /* MeshEnt has no randomize(); the vertex colour is the material's Kd */
            for (int q = 0; q < 3; q++) e->box_colors[b][q] = e->box_base[b] >= 0 ? COLORS[e->box_base[b]][q] : 1.0;
        } else if (ent->kind == MWO_ENT_TEXT) {   /* TextFrame.randomize (entity.py:268-278): Texture.get(name, rng) per character */
            for (int c = 0; c < ent->n_chars; c++)
                if (ent->tex[c] >= 0) ent->tex[c] = ent->tex[c] + (dr ? (int)rs_randint(&e->rng, 0, 9) : 0);
        }
    }
    if (e->n_order < 1 || e->order[e->n_order - 1] != AGENT_SLOT) fail("the agent must be the last entity placed");
    /* Agent.randomize, entity.py:486-492 */
    sample_param(e, dr, MWO_P_CAM_HEIGHT, &e->cam_height);
    sample_param(e, dr, MWO_P_CAM_FWD_DISP, &e->cam_fwd_disp);
    sample_param(e, dr, MWO_P_CAM_PITCH, &e->cam_pitch);
    sample_param(e, dr, MWO_P_CAM_FOV_Y, &e->cam_fov_y);
    if (!e->static_done) gen_static_data(e);
    for (int b = 0; b < MWO_MAX_BOXES; b++) {
        memcpy(e->frame_pos[b], e->boxes[b].pos, sizeof(e->frame_pos[b])); e->frame_dir[b] = e->boxes[b].dir; e->frame_alive[b] = e->boxes[b].alive;
    }
}

/* MiniWorldEnv.near, miniworld.py:961-971 */
static int near_ent(MwoEnv *e, const Ent *b) {
    double d[3] = {b->pos[0] - e->agent.pos[0], b->pos[1] - e->agent.pos[1], b->pos[2] - e->agent.pos[2]};
    double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    int f32;   /* (ent0.radius + ent1.radius) + 1.1 * max_forward_step, float32 all the way once a float32 radius is in it */
    double thr = tagged_add(b->radius, b->rad_f32, e->agent.radius, 0, &f32);
    thr = tagged_add(thr, f32, 1.1 * e->max_forward_step, 0, NULL);
    return dist < thr;
}
static int near_box(MwoEnv *e) { return near_ent(e, &e->boxes[0]); }

/* MiniWorldEnv._get_carry_pos, miniworld.py:594-606 */
static void carry_pos(MwoEnv *e, const double *agent_pos, const Ent *ent, double *out) {
    int f32;
    double dist = tagged_add(e->agent.radius, 0, ent->radius, ent->rad_f32, &f32);   /* self.agent.radius + ent.radius + self.max_forward_step */
    dist = tagged_add(dist, f32, e->max_forward_step, 0, NULL);
    double dv[3] = {cos(e->agent.dir), 0, -sin(e->agent.dir)};
    for (int k = 0; k < 3; k++) out[k] = agent_pos[k] + (dv[k] * 1.05) * dist;   /* agent_pos + self.agent.dir_vec * 1.05 * dist */
    double y_pos = (e->cam_height - ent->height) - 0.3;
    if (!(y_pos > 0)) y_pos = 0;   /* max(cam_height - ent.height - 0.3, 0) */
    out[1] = out[1] + 1.0 * y_pos;  /* pos + Y_VEC * y_pos */
    out[0] = out[0] + 0.0 * y_pos; out[2] = out[2] + 0.0 * y_pos;
}

static void remove_from_order(MwoEnv *e, int slot) {   /* self.entities.remove(ent) */
    int k = 0;
    while (k < e->n_order && e->order[k] != slot) k++;
    if (k == e->n_order) fail("entity not in the list");
    for (; k + 1 < e->n_order; k++) e->order[k] = e->order[k + 1];
    e->n_order--;
}

/* MiniWorldEnv.step miniworld.py:658-716 + task rule (e.g. envs/maze.py:106-113) */
void mwo_step(MwoEnv *e, int action, double *reward, int *done) {
    if (e->task == MWO_SIM2REAL_PUSH && action == 2) { /* simtorealpush.py:109-125, before MiniWorldEnv.step */
        double fwd_dist = e->params[MWO_P_FORWARD_STEP].hi[0];
        double dv[3] = {cos(e->agent.dir), 0, -sin(e->agent.dir)};
        double next_pos[3];
        for (int k = 0; k < 3; k++) next_pos[k] = e->agent.pos[k] + dv[k] * fwd_dist;
        Ent *boxes[2] = {&e->boxes[0], &e->boxes[1]};
        for (int b = 0; b < 2; b++) {
            Ent *box = boxes[b];
            double vec[3] = {box->pos[0] - next_pos[0], box->pos[1] - next_pos[1], box->pos[2] - next_pos[2]};
            double dist = sqrt((vec[0] * vec[0] + vec[1] * vec[1]) + vec[2] * vec[2]);
            if (dist < e->agent.radius + box->radius) {
                double nb[3] = {box->pos[0] + vec[0], box->pos[1] + vec[1], box->pos[2] + vec[2]};
                if (!intersect(e, box, nb, box->radius)) {
                    memcpy(box->pos, nb, sizeof(nb));
                    box->dir += rs_uniform(&e->rng, -M_PI / 5, M_PI / 5);
                }
            }
        }
    }
    e->step_count += 1;
    int dr = e->domain_rand;
    double fwd_step, fwd_drift, turn_step;
    sample_param(e, dr, MWO_P_FORWARD_STEP, &fwd_step);
    sample_param(e, dr, MWO_P_FORWARD_DRIFT, &fwd_drift);
    sample_param(e, dr, MWO_P_TURN_STEP, &turn_step);
    Ent *carried = e->carrying >= 0 ? &e->boxes[e->carrying] : NULL;
    if (action == 2 || action == 3) { /* move_agent miniworld.py:608-633 */
        double fd = action == 2 ? fwd_step : -fwd_step;
        double dv[3] = {cos(e->agent.dir), 0, -sin(e->agent.dir)}; /* entity.py:72-80 */
        double rv[3] = {sin(e->agent.dir), 0, cos(e->agent.dir)};  /* entity.py:82-90 */
        double np_[3];
        for (int k = 0; k < 3; k++) np_[k] = (e->agent.pos[k] + dv[k] * fd) + rv[k] * fwd_drift;
        if (!intersect(e, &e->agent, np_, e->agent.radius)) {
            int ok = 1;
            double cp[3];
            if (carried) { /* the carried entity must fit where it would go (miniworld.py:622-629) */
                carry_pos(e, np_, carried, cp);
                if (intersect_t(e, carried, cp, carried->radius, carried->rad_f32)) ok = 0;
                else memcpy(carried->pos, cp, sizeof(cp));
            }
            if (ok) memcpy(e->agent.pos, np_, sizeof(np_));
        }
    } else if (action == 0 || action == 1) { /* turn_agent miniworld.py:635-656 */
        double ta = action == 0 ? turn_step : -turn_step;
        ta *= (M_PI / 180);
        double orig = e->agent.dir;
        e->agent.dir += ta;
        if (carried) {
            double cp[3];
            carry_pos(e, e->agent.pos, carried, cp);
            if (intersect_t(e, carried, cp, carried->radius, carried->rad_f32)) e->agent.dir = orig;
            else { memcpy(carried->pos, cp, sizeof(cp)); carried->dir = e->agent.dir; }
        }
    } else if (action == 4) { /* pickup, miniworld.py:682-689: the first entity within 1.2 r of a point 1.5 r ahead */
        double dv[3] = {cos(e->agent.dir), 0, -sin(e->agent.dir)};
        double tp[3];
        for (int k = 0; k < 3; k++) tp[k] = e->agent.pos[k] + (dv[k] * 1.5) * e->agent.radius;
        int hit = intersect(e, &e->agent, tp, 1.2 * e->agent.radius);
        if (e->carrying < 0 && hit >= 2 && hit - 2 < e->n_boxes && !e->boxes[hit - 2].is_static) e->carrying = hit - 2; /* `if not ent.is_static` */
    } else if (action == 5) { /* drop, miniworld.py:692-695 */
        if (e->carrying >= 0) { e->boxes[e->carrying].pos[1] = 0; e->carrying = -1; }
    }
    if (e->carrying >= 0) { /* miniworld.py:698-701 */
        Ent *c = &e->boxes[e->carrying];
        double cp[3];
        carry_pos(e, e->agent.pos, c, cp);
        memcpy(c->pos, cp, sizeof(cp));
        c->dir = e->agent.dir;
    }
    for (int b = 0; b < MWO_MAX_BOXES; b++) {   /* obs = self.render_obs() happens here (miniworld.py:705) */
        memcpy(e->frame_pos[b], e->boxes[b].pos, sizeof(e->frame_pos[b])); e->frame_dir[b] = e->boxes[b].dir; e->frame_alive[b] = e->boxes[b].alive;
    }
    double r = 0; int d = 0;
    if (e->step_count >= e->max_episode_steps) { d = 1; r = 0; }
    if (e->task == MWO_TMAZE_TWOBOX) { /* tmaze.py:196-208 / 299-320: goal box, then penalty box */
        const Ent *boxes[2] = {&e->boxes[0], &e->boxes[1]};
        if (near_ent(e, boxes[e->goal_idx])) { r += 1.0 - 0.2 * ((double)e->step_count / e->max_episode_steps); d = 1; }
        if (near_ent(e, boxes[1 - e->goal_idx])) { r += -1 * (1.0 - 0.2 * ((double)e->step_count / e->max_episode_steps)); d = 1; }
        e->feature[0] = e->feature[1] = 0;
        if (e->task_args[0] != 0) { /* feature = [near(blue), near(red)], tmaze.py:311-318 */
            e->feature[0] = near_ent(e, &e->boxes[1]) ? 1 : 0;
            e->feature[1] = near_ent(e, &e->boxes[0]) ? 1 : 0;
            e->task_step_count += 1;
        }
    } else if (e->task == MWO_SIM2REAL_PUSH) { /* simtorealpush.py:129-133 */
        double dx = e->boxes[0].pos[0] - e->boxes[1].pos[0], dy = e->boxes[0].pos[1] - e->boxes[1].pos[1], dz = e->boxes[0].pos[2] - e->boxes[1].pos[2];
        if (sqrt((dx * dx + dy * dy) + dz * dz) < e->goal_dist) { r = 1; d = 1; }
    } else if (e->task == MWO_PUTNEXT) { /* putnext.py:45-53: red (4) next to yellow (5), nothing carried */
        if (e->carrying < 0) {
            const Ent *a = &e->boxes[4], *b = &e->boxes[5];
            double dd[3] = {a->pos[0] - b->pos[0], a->pos[1] - b->pos[1], a->pos[2] - b->pos[2]};
            double dist = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
            if (dist < a->radius + b->radius + 1.1 * e->max_forward_step) { r += 1.0 - 0.2 * ((double)e->step_count / e->max_episode_steps); d = 1; }
        }
    } else if (e->task == MWO_PICKUPOBJS) { /* pickupobjs.py:56-69: a picked object leaves the list (AFTER the frame was rendered) */
        if (e->carrying >= 0) {
            remove_from_order(e, e->carrying);
            e->boxes[e->carrying].alive = 0;
            e->carrying = -1;
            e->num_picked += 1;
            r = 1;
            if (e->num_picked == e->n_boxes) d = 1;
        }
    } else if (e->task == MWO_ROOMOBJS || e->task == MWO_THREEROOMS) { /* roomobjs.py:49-51, threerooms.py:71-74: nothing */
    } else if (e->task == MWO_COLLECTHEALTH) { /* collecthealth.py:51-77 */
        e->health -= 2;
        if (action == 4 && e->carrying >= 0) {   /* respawn the kit: entities.remove(), place_entity() appends it to the END of the list */
            Ent *kit = &e->boxes[e->carrying];
            remove_from_order(e, e->carrying);
            place_entity(e, kit, 0, 0, NOVAL, NOVAL, NOVAL, NOVAL);
            e->carrying = -1;
            e->health = 100;
        }
        if (e->health > 0) r = 2;
        else { r = -100; d = 1; }
    } else if (e->task == MWO_SIGN) { /* sign.py:115-128 */
        if (action == 3) d = 1;   /* move_forward + 1: the custom end-of-episode action (the base class has moved the agent back) */
        for (int obj = 0; obj < 2; obj++)
            for (int ci = 0; ci < 3; ci++)
                if (near_ent(e, &e->boxes[obj * 3 + ci])) { d = 1; r = (double)(ci == (int)e->task_args[1] && obj == (int)e->task_args[2]) * 2 - 1; }
    } else if (e->task == MWO_SIDEWALK) { /* sidewalk.py:74-87 */
        if (point_inside(&e->rooms[1], e->agent.pos)) { r = 0; d = 1; }   /* walking into the street ends the episode */
        if (near_ent(e, &e->boxes[6])) { r += 1.0 - 0.2 * ((double)e->step_count / e->max_episode_steps); d = 1; }
    } else if (near_box(e)) { r += 1.0 - 0.2 * ((double)e->step_count / e->max_episode_steps); d = 1; }
    *reward = r; *done = d;
}

/* math.py:9-23 */
void mwo_gen_rot_matrix(const double *axis_in, double angle, double *m) {
    double n = sqrt(axis_in[0] * axis_in[0] + axis_in[1] * axis_in[1] + axis_in[2] * axis_in[2]);
    double axis[3] = {axis_in[0] / n, axis_in[1] / n, axis_in[2] / n};
    double a = cos(angle / 2.0), s = sin(angle / 2.0);
    double b = -axis[0] * s, c = -axis[1] * s, d = -axis[2] * s;
    m[0] = a * a + b * b - c * c - d * d; m[1] = 2 * (b * c - a * d); m[2] = 2 * (b * d + a * c);
    m[3] = 2 * (b * c + a * d); m[4] = a * a + c * c - b * b - d * d; m[5] = 2 * (c * d - a * b);
    m[6] = 2 * (b * d - a * c); m[7] = 2 * (c * d + a * b); m[8] = a * a + d * d - b * b - c * c;
}

#ifndef MAT_ROW_DOT_EXPR
#define MAT_ROW_DOT_EXPR ((v[0] * m[j] + v[1] * m[3 + j]) + v[2] * m[6 + j])
#endif
static void mat_row_dot(const double *v, const double *m, double *out) {
    for (int j = 0; j < 3; j++) out[j] = MAT_ROW_DOT_EXPR;
}

static void vec_mat(const double *v, const double *m, double *out) { /* np.dot(v, M) */
    for (int j = 0; j < 3; j++) out[j] = (v[0] * m[0 * 3 + j] + v[1] * m[1 * 3 + j]) + v[2] * m[2 * 3 + j];
}

/* Agent.cam_pos / cam_dir, entity.py:457-484 */
static void camera(MwoEnv *e, double *cam_pos, double *cam_dir) {
    const double Yv[3] = {0, 1, 0}, Zv[3] = {0, 0, 1}, Xv[3] = {1, 0, 0};
    double rot_y[9], rot_z[9], disp[3], t[3];
    mwo_gen_rot_matrix(Yv, e->agent.dir, rot_y);
    double cd[3] = {e->cam_fwd_disp, e->cam_height, 0};
    vec_mat(cd, rot_y, disp);
    for (int k = 0; k < 3; k++) cam_pos[k] = e->agent.pos[k] + disp[k];
    mwo_gen_rot_matrix(Zv, e->cam_pitch * M_PI / 180, rot_z);
    vec_mat(Xv, rot_z, t);
    vec_mat(t, rot_y, cam_dir);
}

void mwo_get_state(MwoEnv *e, MwoState *s) {
    memset(s, 0, sizeof(*s));
    memcpy(s->agent_pos, e->agent.pos, 24); s->agent_dir = e->agent.dir;
    memcpy(s->box_pos, e->boxes[0].pos, 24); s->box_dir = e->boxes[0].dir; memcpy(s->box_color, e->box_colors[0], 24);
    s->cam_height = e->cam_height; s->cam_fwd_disp = e->cam_fwd_disp; s->cam_pitch = e->cam_pitch; s->cam_fov_y = e->cam_fov_y;
    memcpy(s->sky_color, e->sky_color, 24); memcpy(s->light_pos, e->light_pos, 24);
    memcpy(s->light_color, e->light_color, 24); memcpy(s->light_ambient, e->light_ambient, 24);
    camera(e, s->cam_pos, s->cam_dir);
    s->step_count = e->step_count; s->max_episode_steps = e->max_episode_steps;
    s->n_rooms = e->n_rooms; s->n_segs = e->n_segs;
    int nq = 0; for (int i = 0; i < e->n_rooms; i++) nq += e->rooms[i].n_quads;
    s->n_quads = nq;
    s->rng_pos = e->rng.pos; s->rng_key0 = e->rng.key[0]; s->rng_key1 = e->rng.key[1]; s->rng_key623 = e->rng.key[623];
    uint64_t sum = 0; for (int i = 0; i < 624; i++) sum += e->rng.key[i];
    s->rng_keysum = (uint32_t)(sum & 0xFFFFFFFFu);
    s->n_boxes = e->n_boxes; s->goal_idx = e->goal_idx;
    memcpy(s->box2_pos, e->boxes[1].pos, 24); s->box2_dir = e->boxes[1].dir; memcpy(s->box2_color, e->box_colors[1], 24);
    s->episode_count = e->episode_count; s->task_step_count = e->task_step_count;
    s->feature[0] = e->feature[0]; s->feature[1] = e->feature[1];
    s->box_size = e->box_s[0]; s->box2_size = e->box_s[1]; s->agent_radius = e->agent.radius; s->goal_dist = e->goal_dist;
    for (int b = 0; b < MWO_MAX_BOXES; b++) {
        memcpy(s->boxes_pos[b], e->boxes[b].pos, 24); s->boxes_dir[b] = e->boxes[b].dir;
        memcpy(s->boxes_color[b], e->box_colors[b], 24); s->boxes_size[b] = b < e->n_boxes ? e->box_s[b] : 0;
    }
    s->carrying = e->carrying;
    for (int b = 0; b < MWO_MAX_BOXES; b++) {
        const Ent *t = &e->boxes[b];
        s->ents_kind[b] = t->kind; s->ents_mesh[b] = t->geom; s->ents_alive[b] = b < e->n_boxes ? t->alive : 0; s->ents_static[b] = t->is_static;
        s->ents_rad_f32[b] = t->rad_f32; s->ents_radius[b] = t->radius; s->ents_height[b] = t->height; s->ents_scale[b] = t->scale;
        for (int c = 0; c < 8; c++) s->ents_tex[b][c] = t->kind == MWO_ENT_TEXT ? (c < t->n_chars ? t->tex[c] : -1) : (c == 0 && t->kind == MWO_ENT_IMAGE ? t->tex[0] : -1);
    }
    for (int k = 0; k <= MWO_MAX_BOXES; k++) s->order[k] = k < e->n_order ? e->order[k] : -1;
    s->n_order = e->n_order; s->health = e->health; s->num_picked = e->num_picked;
}
void mwo_set_counters(MwoEnv *e, long long episode_count, long long task_step_count, int goal_idx) {
    e->episode_count = episode_count; e->task_step_count = task_step_count; e->goal_idx = goal_idx;
}

void mwo_set_agent(MwoEnv *e, double x, double z, double dir) { e->agent.pos[0] = x; e->agent.pos[1] = 0; e->agent.pos[2] = z; e->agent.dir = dir; }
void mwo_set_step_count(MwoEnv *e, int sc) { e->step_count = sc; }
void mwo_set_box(MwoEnv *e, int b, double x, double z, double dir) {
    Ent *t = &e->boxes[b];
    t->pos[0] = x; t->pos[1] = 0; t->pos[2] = z; t->dir = dir;
}
void mwo_set_box_y(MwoEnv *e, int b, double y) { e->boxes[b].pos[1] = y; }
void mwo_set_carrying(MwoEnv *e, int b) { e->carrying = b; }
/* 1: mwo_render draws the entities as the last step's own frame saw them (PickupObjs removes, CollectHealth respawns an entity
 * AFTER the frame was rendered, pickupobjs.py:56-69, collecthealth.py:51-64); 0: the current state */
void mwo_render_step_frame(MwoEnv *e, int on) { e->render_step_frame = on; }

void mwo_get_geometry(MwoEnv *e, double *outline, double *heights, double *portals, int *portal_count, double *segs,
                      double *room_probs, double *qv, double *qn, float *qt, int *qoff, double *ftex, double *ctex, int *tex_ids) {
    int q = 0;
    for (int i = 0; i < e->n_rooms; i++) {
        Room *r = &e->rooms[i];
        for (int k = 0; k < 4; k++) { outline[(i * 4 + k) * 2] = k < r->n_edges ? r->outline[k][0] : NAN; outline[(i * 4 + k) * 2 + 1] = k < r->n_edges ? r->outline[k][2] : NAN; }
        heights[i] = r->wall_height;
        for (int ed = 0; ed < 4; ed++) {
            portal_count[i * 4 + ed] = r->n_portals[ed];
            for (int k = 0; k < MWO_MAX_PORTALS; k++) {
                double *o = portals + ((i * 4 + ed) * MWO_MAX_PORTALS + k) * 4;
                if (k < r->n_portals[ed]) { o[0] = r->portals[ed][k].start; o[1] = r->portals[ed][k].end; o[2] = r->portals[ed][k].min_y; o[3] = r->portals[ed][k].max_y; }
                else o[0] = o[1] = o[2] = o[3] = NAN;
            }
        }
        room_probs[i] = e->room_probs[i];
        qoff[i] = q;
        for (int k = 0; k < r->n_quads; k++, q++)
            for (int v = 0; v < 4; v++) {
                for (int c = 0; c < 3; c++) { qv[(q * 4 + v) * 3 + c] = r->quads[k].verts[v][c]; qn[(q * 4 + v) * 3 + c] = r->quads[k].norm[c]; }
                qt[(q * 4 + v) * 2] = r->quads[k].texcs[v][0]; qt[(q * 4 + v) * 2 + 1] = r->quads[k].texcs[v][1];
            }
        for (int v = 0; v < 4; v++) {
            const int in = v < r->n_edges;
            ftex[(i * 4 + v) * 2] = in ? r->floor_texcs[v][0] : NAN; ftex[(i * 4 + v) * 2 + 1] = in ? r->floor_texcs[v][1] : NAN;
            ctex[(i * 4 + v) * 2] = in ? r->ceil_texcs[v][0] : NAN; ctex[(i * 4 + v) * 2 + 1] = in ? r->ceil_texcs[v][1] : NAN;
        }
        tex_ids[i * 3] = r->wall_tex; tex_ids[i * 3 + 1] = r->floor_tex; tex_ids[i * 3 + 2] = r->ceil_tex;
    }
    qoff[e->n_rooms] = q;
    for (int s = 0; s < e->n_segs; s++) {
        segs[s * 4] = e->wall_segs[s][0][0]; segs[s * 4 + 1] = e->wall_segs[s][0][2];
        segs[s * 4 + 2] = e->wall_segs[s][1][0]; segs[s * 4 + 3] = e->wall_segs[s][1][2];
    }
}

/* ===================================================================== render (float32) */
/* Restates what the reference asks of OpenGL (render_obs miniworld.py:1160-1205,
 * _render_static 1014-1057, Room._render 390-423, Box.render entity.py:385-408,
 * drawBox opengl.py:394-444, Texture.load opengl.py:85-103, FrameBuffer.resolve 283-334,
 * get_depth_map 336-371).  Frozen choices where GL leaves freedom:
 *   - 8 samples/pixel at the standard 8x pattern; coverage + nearest depth per sample,
 *     exact (unquantised) depth ordering, ties -> earlier-drawn surface;
 *   - one shade per (pixel, surface) at the pixel centre, plane extrapolated;
 *   - LOD from forward differences of the surface's texcoords at the pixel centre,
 *     trilinear REPEAT filtering over a 2x2-box-filtered RGBA8 mip chain;
 *   - resolve = mean of the 8 sample colours, unorm8 round-to-nearest;
 *   - depth resolve takes sample 0, DEPTH16 round-to-nearest.
 * Visibility uses portal traversal of the axis-aligned rooms (watertight by construction);
 * tests/ cross-check it against a brute-force z-buffer over the reference's own polygon stream.
 */
typedef struct {
    float lo, hi, min_y, max_y; /* portal extent along the side's axis (world coord) and in y */
    int nbr;                    /* room behind the portal, -1 none */
    float u_org, u_sgn;         /* wall texcoord: s = (coord - u_org) * u_sgn * scale */
} RSide;
typedef struct {
    float min_x, max_x, min_z, max_z, height;
    int wall_tex, floor_tex, ceil_tex, no_ceiling;
    RSide side[4]; /* 0:+x (east) 1:-z (north) 2:-x (west) 3:+z (south) */
} RRoom;

static const float SAMPLE_X[8] = {1, -1, 5, -3, -5, -7, 3, 7};
static const float SAMPLE_Y[8] = {-3, 3, 1, -5, 5, -1, 7, -7};

typedef struct {
    float eye[3], F[3], S[3], U[3], TW, TH;
    int W, H;
    float invW, invH;
} Cam;

static void make_ray(const Cam *c, float wx, float wy, float *d) {
    float nx = (2.0f * wx - (float)c->W) * c->invW, ny = (2.0f * wy - (float)c->H) * c->invH;
    float ax = nx * c->TW, ay = ny * c->TH;
    for (int k = 0; k < 3; k++) d[k] = fmaf(c->U[k], ay, fmaf(c->S[k], ax, c->F[k]));
}

typedef struct { int kind; /* 0 sky, 1 floor, 2 ceil, 3 wall, 4 box */ int room, side; float t; } Hit;

/* A wall with TWO portals (ThreeRooms: the big room's south wall opens into both small rooms) has no place in a room record with
 * one portal per side: the room is cut in two RENDER rooms along the line midway between the two openings, joined by a "virtual"
 * portal as wide and as high as the cut (nothing is drawn there).  The cut leaves visibility unchanged - a convex room cut by a
 * plane is two convex rooms - and every surface keeps its texture origin.  The lower half (smaller coordinate along the wall) keeps
 * the room's index, the upper half is appended behind the last room.  Returns the number of render rooms (<= 2 n). */
static int build_rrooms(MwoEnv *e, RRoom *rr) {
    int n = e->n_rooms, n_out = e->n_rooms;
    for (int i = 0; i < n; i++) {
        Room *r = &e->rooms[i]; RRoom *o = &rr[i];
        int cut_s = -1, cut_ax = 0;
        RSide cut_second;
        memset(&cut_second, 0, sizeof(cut_second));
        o->min_x = (float)r->min_x; o->max_x = (float)r->max_x; o->min_z = (float)r->min_z; o->max_z = (float)r->max_z;
        o->height = (float)r->wall_height;
        o->wall_tex = r->wall_tex; o->floor_tex = r->floor_tex; o->ceil_tex = r->ceil_tex; o->no_ceiling = r->no_ceiling;
        for (int s = 0; s < 4; s++) { o->side[s].nbr = -1; o->side[s].lo = o->side[s].hi = o->side[s].min_y = o->side[s].max_y = 0; }
        for (int ed = 0; ed < 4; ed++) {
            const double *nrm = r->edge_norms[ed];
            int s;
            if (nrm[0] == -1 && nrm[2] == 0) s = 0; else if (nrm[0] == 0 && nrm[2] == 1) s = 1;
            else if (nrm[0] == 1 && nrm[2] == 0) s = 2; else if (nrm[0] == 0 && nrm[2] == -1) s = 3;
            else fail("render: only axis-aligned rectangular rooms are supported");
            int ax = (s == 0 || s == 2) ? 2 : 0; /* coordinate that runs along this side */
            double p0c = r->outline[ed][ax], dirc = r->edge_dirs[ed][ax];
            o->side[s].u_org = (float)p0c; o->side[s].u_sgn = (float)dirc;
            if (r->n_portals[ed] > 2) fail("render: at most two portals per edge");
            if (r->n_portals[ed] >= 1) {
                Portal *p = &r->portals[ed][0];
                double c0 = p0c + dirc * p->start, c1 = p0c + dirc * p->end;
                o->side[s].lo = (float)(c0 < c1 ? c0 : c1); o->side[s].hi = (float)(c0 < c1 ? c1 : c0);
                o->side[s].min_y = (float)p->min_y; o->side[s].max_y = (float)p->max_y;
                o->side[s].nbr = -2; /* resolved below */
            }
            if (r->n_portals[ed] == 2) {   /* the second opening: the room is cut between the two (below) */
                Portal *q = &r->portals[ed][1];
                double d0 = p0c + dirc * q->start, d1 = p0c + dirc * q->end;
                if (cut_s >= 0) fail("render: two walls with two portals each");
                cut_s = s; cut_ax = ax;
                cut_second = o->side[s];
                cut_second.lo = (float)(d0 < d1 ? d0 : d1); cut_second.hi = (float)(d0 < d1 ? d1 : d0);
                cut_second.min_y = (float)q->min_y; cut_second.max_y = (float)q->max_y;
            }
        }
        if (cut_s >= 0) {
            const int s = cut_s, ax = cut_ax;
            if (r->n_portals[0] + r->n_portals[1] + r->n_portals[2] + r->n_portals[3] > 2) fail("render: a cut room may have portals on the cut wall only");
            RSide first = o->side[s], second = cut_second;
            if (second.lo < first.lo) { RSide t = first; first = second; second = t; }
            const float cut = (float)(0.5 * ((double)first.hi + (double)second.lo));
            RRoom *hi_half = &rr[n_out];
            *hi_half = *o;
            const int lo_end = ax == 0 ? 2 : 1, hi_end = ax == 0 ? 0 : 3;   /* the sides at the low / high end of the wall's axis: -x / +x, -z (side 1) / +z (side 3) */
            if (ax == 0) { o->max_x = cut; hi_half->min_x = cut; } else { o->max_z = cut; hi_half->min_z = cut; }
            o->side[s] = first; hi_half->side[s] = second;
            RSide v;   /* the virtual portal: the whole cut, from the floor up */
            v.lo = (ax == 0 ? o->min_z : o->min_x) - 1.0f; v.hi = (ax == 0 ? o->max_z : o->max_x) + 1.0f; v.min_y = 0.0f; v.max_y = 1e30f;
            v.u_org = 0.0f; v.u_sgn = 1.0f;
            v.nbr = n_out; o->side[hi_end] = v;
            v.nbr = i; hi_half->side[lo_end] = v;
            n_out++;
        }
    }
    n = n_out;
    /* geometric neighbour inference: the room whose opposite side lies on the same plane with a matching portal */
    for (int i = 0; i < n; i++)
        for (int s = 0; s < 4; s++) {
            if (rr[i].side[s].nbr != -2) continue;
            float plane = s == 0 ? rr[i].max_x : s == 1 ? rr[i].min_z : s == 2 ? rr[i].min_x : rr[i].max_z;
            int os = (s + 2) % 4, found = -1;
            for (int j = 0; j < n && found < 0; j++) {
                if (j == i || rr[j].side[os].nbr == -1) continue;
                float pj = os == 0 ? rr[j].max_x : os == 1 ? rr[j].min_z : os == 2 ? rr[j].min_x : rr[j].max_z;
                if (fabsf(pj - plane) < 1e-4f && fabsf(rr[j].side[os].lo - rr[i].side[s].lo) < 1e-4f &&
                    fabsf(rr[j].side[os].hi - rr[i].side[s].hi) < 1e-4f) found = j;
            }
            if (found < 0) fail("render: portal without a neighbour room");
            rr[i].side[s].nbr = found;
        }
    return n;
}

static Hit trace_rooms(const RRoom *rr, int n_rooms, int room, const float *o, const float *d) {
    Hit h = {0, -1, -1, INFINITY};
    if (room < 0) return h;
    /* per-ray reciprocals (one correctly rounded division each); plane distances are (c - o) * inv */
    float ix = d[0] != 0 ? 1.0f / d[0] : 0.0f, iy = d[1] != 0 ? 1.0f / d[1] : 0.0f, iz = d[2] != 0 ? 1.0f / d[2] : 0.0f;
    for (int iter = 0; iter < n_rooms + 1; iter++) {
        const RRoom *r = &rr[room];
        float tx = INFINITY, tz = INFINITY; int sx = 0, sz = 1;
        if (d[0] > 0) { tx = (r->max_x - o[0]) * ix; sx = 0; } else if (d[0] < 0) { tx = (r->min_x - o[0]) * ix; sx = 2; }
        if (d[2] > 0) { tz = (r->max_z - o[2]) * iz; sz = 3; } else if (d[2] < 0) { tz = (r->min_z - o[2]) * iz; sz = 1; }
        float ts; int s;
        if (tx <= tz) { ts = tx; s = sx; } else { ts = tz; s = sz; }
        if (d[1] < 0) { float tf = (0.0f - o[1]) * iy; if (tf <= ts) { h.kind = 1; h.room = room; h.t = tf; return h; } }
        if (d[1] > 0) { /* a room without ceiling polygon (Room._render, miniworld.py:406): crossing its ceiling plane - which
                         * every ray passing over a wall's top edge does first - means sky */
            float tc = (r->height - o[1]) * iy;
            if (tc <= ts) { if (r->no_ceiling) return h; h.kind = 2; h.room = room; h.t = tc; return h; }
        }
        if (!(ts < INFINITY)) return h;
        const RSide *sd = &r->side[s];
        float hc = (s == 0 || s == 2) ? fmaf(ts, d[2], o[2]) : fmaf(ts, d[0], o[0]);
        float y = fmaf(ts, d[1], o[1]);
        if (sd->nbr >= 0 && sd->lo < hc && hc < sd->hi && sd->min_y < y && y < sd->max_y) { room = sd->nbr; continue; }
        h.kind = 3; h.room = room; h.side = s; h.t = ts; return h;
    }
    return h;
}

/* ---- rooms that are convex polygons with 3 or 4 arbitrary edges (YMaze: a triangular hub, arms rotated by 120 degrees,
 * sliver-shaped connectors; envs/ymaze.py:28-83).  Same spec as above with the slab distances of an axis-aligned rectangle
 * replaced by ray / edge-plane distances: for edge i with start p, unit direction dir and inward unit normal n (float32 of the
 * state half's float64 values), a ray (o, d) leaves through it if den = fmaf(n.z, d.z, n.x d.x) < 0, at
 * t = fmaf(n.z, p.z - o.z, n.x (p.x - o.x)) / den (one correctly rounded division); the exit edge is the first one, in edge
 * order, with the smallest t; the crossing point's distance along the edge is hc = fmaf(dir.z, hz - p.z, dir.x (hx - p.x)) with
 * (hx, hz) = fmaf(t, d, o), which is also the wall texture's u coordinate (Room._gen_static_data: seg_start runs from edge_p0).
 * A room whose outline runs the other way round (the connector rooms connect_rooms builds between YMaze's hub and arms do:
 * miniworld.py:826 lists their corners clockwise there) has all its polygons facing away from its inside - floor downwards,
 * ceiling upwards, walls outwards - so with back-face culling on (miniworld.py:498-499) nothing of it is drawn for a viewer
 * inside the world.  Geometrically such a connector has NEGATIVE width - YMaze's arms overlap the hub by a centimetre, the
 * connector spans the overlap back to front - so the traversal by-passes it: a portal that leads into a culled connector leads
 * on to the room behind the connector's other portal, and a culled room never holds the eye. */
typedef struct { float px, pz, dx, dz, nx, nz, lo, hi, max_y; int nbr; } PEdge;
typedef struct { int ne; float height; int wall_tex, floor_tex, ceil_tex, no_ceiling, culled; PEdge e[4]; } PRoom;

static int build_prooms(MwoEnv *e, PRoom *pr) {
    for (int i = 0; i < e->n_rooms; i++) {
        Room *r = &e->rooms[i]; PRoom *o = &pr[i];
        o->ne = r->n_edges; o->height = (float)r->wall_height;
        o->wall_tex = r->wall_tex; o->floor_tex = r->floor_tex; o->ceil_tex = r->ceil_tex; o->no_ceiling = r->no_ceiling;
        double cx = 0, cz = 0, inward = 0;
        for (int k = 0; k < r->n_edges; k++) { cx += r->outline[k][0] / r->n_edges; cz += r->outline[k][2] / r->n_edges; }
        for (int k = 0; k < r->n_edges; k++) inward += r->edge_norms[k][0] * (cx - r->outline[k][0]) + r->edge_norms[k][2] * (cz - r->outline[k][2]);
        o->culled = inward < 0;   /* "inward" normals point away from the centroid: reversed winding */
        for (int k = 0; k < r->n_edges; k++) {
            PEdge *pe = &o->e[k];
            pe->px = (float)r->outline[k][0]; pe->pz = (float)r->outline[k][2];
            pe->dx = (float)r->edge_dirs[k][0]; pe->dz = (float)r->edge_dirs[k][2];
            pe->nx = (float)r->edge_norms[k][0]; pe->nz = (float)r->edge_norms[k][2];
            pe->lo = pe->hi = pe->max_y = 0; pe->nbr = -1;
            if (r->n_portals[k] > 1) fail("render: at most one portal per edge");
            if (r->n_portals[k] == 1) {
                if (r->portals[k][0].min_y != 0) fail("render: portals start at the floor");
                pe->lo = (float)r->portals[k][0].start; pe->hi = (float)r->portals[k][0].end; pe->max_y = (float)r->portals[k][0].max_y;
                pe->nbr = r->nbr[k];
                if (pe->nbr < 0) fail("render: portal without a neighbour room");
            }
        }
    }
    for (int i = 0; i < e->n_rooms; i++)   /* by-pass culled connectors */
        for (int k = 0; k < pr[i].ne; k++) {
            int c = pr[i].e[k].nbr;
            if (c < 0 || !pr[c].culled || pr[i].culled) continue;
            int onward = -1;
            for (int q = 0; q < pr[c].ne; q++)
                if (e->rooms[c].n_portals[q] == 1 && e->rooms[c].nbr[q] != i) onward = e->rooms[c].nbr[q];
            if (onward < 0 || pr[onward].culled) fail("render: a culled connector must lead on to a regular room");
            pr[i].e[k].nbr = onward;
        }
    return e->n_rooms;
}

static Hit trace_rooms_poly(const PRoom *pr, int n_rooms, int room, const float *o, const float *d) {
    Hit h = {0, -1, -1, INFINITY};
    if (room < 0) return h;
    float iy = d[1] != 0 ? 1.0f / d[1] : 0.0f;
    for (int iter = 0; iter < n_rooms + 1; iter++) {
        const PRoom *r = &pr[room];
        float ts = INFINITY; int s = -1;
        for (int k = 0; k < r->ne; k++) {
            const PEdge *pe = &r->e[k];
            float den = fmaf(pe->nz, d[2], pe->nx * d[0]);
            if (den < 0) {
                float num = fmaf(pe->nz, pe->pz - o[2], pe->nx * (pe->px - o[0]));
                float t = num / den;
                if (t < ts) { ts = t; s = k; }
            }
        }
        if (d[1] < 0) { float tf = (0.0f - o[1]) * iy; if (tf <= ts) { if (r->culled) return h; h.kind = 1; h.room = room; h.t = tf; return h; } }
        if (d[1] > 0) {
            float tc = (r->height - o[1]) * iy;
            if (tc <= ts) { if (r->no_ceiling || r->culled) return h; h.kind = 2; h.room = room; h.t = tc; return h; }
        }
        if (!(ts < INFINITY)) return h;
        const PEdge *pe = &r->e[s];
        float hx = fmaf(ts, d[0], o[0]), hz = fmaf(ts, d[2], o[2]);
        float hc = fmaf(pe->dz, hz - pe->pz, pe->dx * (hx - pe->px));
        float y = fmaf(ts, d[1], o[1]);
        if (pe->nbr >= 0 && pe->lo < hc && hc < pe->hi && 0.0f < y && y < pe->max_y) { room = pe->nbr; continue; }
        if (r->culled) return h;
        h.kind = 3; h.room = room; h.side = s; h.t = ts; return h;
    }
    return h;
}

static int surf_texcoord_poly(const PRoom *pr, const Hit *h, const float *o, const float *d, float *s, float *t) {
    const PRoom *r = &pr[h->room];
    float tt;
    if (h->kind == 1 || h->kind == 2) {
        float py = h->kind == 1 ? 0.0f : r->height;
        if (d[1] == 0) return 0;
        tt = (py - o[1]) / d[1];
        if (!(tt > 0)) return 0;
        int tex = h->kind == 1 ? r->floor_tex : r->ceil_tex;
        float sc_s = (float)(512.0 / tex_width(tex)), sc_t = (float)(512.0 / tex_height(tex));
        *s = (o[0] + tt * d[0]) * sc_s; *t = (o[2] + tt * d[2]) * sc_t;
        return 1;
    }
    const PEdge *pe = &r->e[h->side];
    float sc_s = (float)(512.0 / tex_width(r->wall_tex)), sc_t = (float)(512.0 / tex_height(r->wall_tex));
    float den = fmaf(pe->nz, d[2], pe->nx * d[0]);
    if (den == 0) return 0;
    tt = fmaf(pe->nz, pe->pz - o[2], pe->nx * (pe->px - o[0])) / den;
    if (!(tt > 0)) return 0;
    float hx = o[0] + tt * d[0], hz = o[2] + tt * d[2], y = o[1] + tt * d[1];
    float hc = fmaf(pe->dz, hz - pe->pz, pe->dx * (hx - pe->px));
    *s = hc * sc_s; *t = y * sc_t;
    return 1;
}

typedef struct { float pos[3], c, s, half[3], sy; } RBox;

/* returns face index 0..5 (-x,+x,-y,+y,-z,+z in box-local axes) or -1; *t_out = entry t */
static int trace_box(const RBox *b, const float *o, const float *d, float *t_out) {
    float ro[3] = {o[0] - b->pos[0], o[1] - b->pos[1], o[2] - b->pos[2]};
    float lo_[3] = {ro[0] * b->c - ro[2] * b->s, ro[1], ro[0] * b->s + ro[2] * b->c};
    float ld[3] = {fmaf(d[0], b->c, -(d[2] * b->s)), d[1], fmaf(d[0], b->s, d[2] * b->c)};
    float lo[3] = {-b->half[0], 0.0f, -b->half[2]}, hi[3] = {b->half[0], b->sy, b->half[2]};
    float tn = -INFINITY, tf = INFINITY; int face = -1;
    for (int a = 0; a < 3; a++) {
        if (ld[a] == 0) { if (lo_[a] < lo[a] || lo_[a] > hi[a]) return -1; continue; }
        float inv = 1.0f / ld[a];
        float t1 = (lo[a] - lo_[a]) * inv, t2 = (hi[a] - lo_[a]) * inv;
        float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
        if (tmin > tn) { tn = tmin; face = a * 2 + (ld[a] > 0 ? 0 : 1); }
        if (tmax < tf) tf = tmax;
    }
    if (face < 0 || !(tn <= tf) || !(tn > 0)) return -1;
    *t_out = tn;
    return face;
}

/* ---- mesh entities (MeshEnt.render, entity.py:130-141: glTranslatef(pos) glScalef(scale) glRotatef(dir) mesh.render()) --------
 * Spec ("meshes"): a sample ray is taken into the mesh's frame - origin lo = R^T (eye - pos) / scale, direction
 * ld = R^T d / scale, so that the ray parameter stays the world's - and meets front-facing triangles only (GL_CULL_FACE is on,
 * miniworld.py:498-499): Moeller-Trumbore in float32 with the products written out, e1 = v1 - v0, e2 = v2 - v0 in float32;
 * det = e1 . (ld x e2) > 0, 0 <= u <= det, v >= 0, u + v <= det, t = (e2 . q) / det > 0; the nearest t wins, the lower triangle
 * index on a tie (GL_LESS: the first one drawn). */
typedef struct {
    float pos[3], c, s, inv_s;     /* translation, cos / sin of the heading, 1 / scale */
    float lo[3];                   /* the eye in the mesh's frame */
    float Ll[3];                   /* the light direction in the mesh's frame, divided by the scale: n . Ll = (R n / scale) . L */
    float kd[3];                   /* vertex colour: the material's Kd */
    const Mesh *m;
    double gate_c[3], gate_r2;     /* conservative bounding sphere (world), for the oracle's own speed only */
} RMesh;

static void cross3f(const float *a, const float *b, float *o) {
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1])); o[1] = fmaf(a[2], b[0], -(a[0] * b[2])); o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
static float dot3f(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

static void mesh_local_dir(const RMesh *rm, const float *d, float *ld) {
    ld[0] = fmaf(d[0], rm->c, -(d[2] * rm->s)) * rm->inv_s; ld[1] = d[1] * rm->inv_s; ld[2] = fmaf(d[0], rm->s, d[2] * rm->c) * rm->inv_s;
}
/* one triangle: 1 if the ray (rm->lo, ld) meets its front; ub, vb = u, v (NOT divided by det), det, t */
static int mesh_tri(const RMesh *rm, int i, const float *ld, float *t_out, float *u_out, float *v_out, float *det_out, int need_inside) {
    const Mesh *m = rm->m;
    const float *v0 = m->verts + i * 9, *e1 = m->e1 + i * 3, *e2 = m->e2 + i * 3;
    float pv[3], qv[3];
    cross3f(ld, e2, pv);
    const float det = dot3f(e1, pv);
    *det_out = det;
    if (!(det > 0.0f)) return 0;
    const float tv[3] = {rm->lo[0] - v0[0], rm->lo[1] - v0[1], rm->lo[2] - v0[2]};
    const float u = dot3f(tv, pv);
    cross3f(tv, e1, qv);
    const float v = dot3f(ld, qv);
    *u_out = u; *v_out = v;
    if (need_inside && (u < 0.0f || u > det || v < 0.0f || u + v > det)) return 0;
    const float t = dot3f(e2, qv) / det;
    *t_out = t;
    return need_inside ? t > 0.0f : 1;
}
/* nearest front-facing triangle along (eye, d): its index or -1 */
static int trace_mesh(const RMesh *rm, const float *o, const float *d, float *t_out) {
    {   /* conservative sphere gate in float64 (never rejects a ray that meets the mesh) */
        double oc[3] = {rm->gate_c[0] - o[0], rm->gate_c[1] - o[1], rm->gate_c[2] - o[2]};
        double oc2 = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2], b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
        double dd = (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2];
        if (oc2 > rm->gate_r2 && !(b > 0 && b * b >= dd * (oc2 - rm->gate_r2))) return -1;
    }
    float ld[3];
    mesh_local_dir(rm, d, ld);
    int best = -1;
    float bt = INFINITY;
    for (int i = 0; i < rm->m->n_tris; i++) {
        float t, u, v, det;
        if (mesh_tri(rm, i, ld, &t, &u, &v, &det, 1) && t < bt) { bt = t; best = i; }
    }
    *t_out = bt;
    return best;
}

static void fetch_texel(const Tex *t, int l, int i, int j, float *rgb) {
    const uint8_t *p = t->data + t->level_off[l] + ((size_t)j * t->lw[l] + i) * 4;
    rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
}

static int imod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

static void bilinear(const Tex *t, int l, float s, float tt, float *rgb) {
    int w = t->lw[l], h = t->lh[l];
    float uu = s * (float)w - 0.5f, vv = tt * (float)h - 0.5f;
    float fu = floorf(uu), fv = floorf(vv);
    float a = uu - fu, b = vv - fv;
    int i0 = imod((int)fu, w), i1 = imod((int)fu + 1, w), j0 = imod((int)fv, h), j1 = imod((int)fv + 1, h);
    float t00[3], t10[3], t01[3], t11[3];
    fetch_texel(t, l, i0, j0, t00); fetch_texel(t, l, i1, j0, t10); fetch_texel(t, l, i0, j1, t01); fetch_texel(t, l, i1, j1, t11);
    for (int k = 0; k < 3; k++) {
        float top = t00[k] * (1.0f - a) + t10[k] * a, bot = t01[k] * (1.0f - a) + t11[k] * a;
        rgb[k] = top * (1.0f - b) + bot * b;
    }
}

/* s,t: texture coordinates at the pixel centre; (sx,tx),(sy,ty): at the +1 pixel neighbours */
static void sample_texture(int tex_id, float s, float t, float sx, float tx, float sy, float ty, int valid, float *rgb) {
    const Tex *T = &g_tex[tex_id];
    if (!T->data) { rgb[0] = rgb[1] = rgb[2] = 255.0f; return; }
    float lambda;
    if (!valid) lambda = (float)(T->n_levels - 1);
    else {
        float dsdx = (sx - s) * (float)T->w, dtdx = (tx - t) * (float)T->h;
        float dsdy = (sy - s) * (float)T->w, dtdy = (ty - t) * (float)T->h;
        float r1 = dsdx * dsdx + dtdx * dtdx, r2 = dsdy * dsdy + dtdy * dtdy;
        float rho2 = r1 > r2 ? r1 : r2;
        if (!(rho2 < INFINITY)) lambda = (float)(T->n_levels - 1);
        else if (rho2 <= 1.0f) lambda = 0.0f;
        else lambda = 0.5f * log2f(rho2);
    }
    float ws = s - floorf(s), wt = t - floorf(t); /* REPEAT */
    if (!(ws >= 0.0f && ws < 1.0f)) ws = 0.0f; /* non-finite coordinates must not become texel addresses */
    if (!(wt >= 0.0f && wt < 1.0f)) wt = 0.0f;
    int maxl = T->n_levels - 1;
    if (lambda <= 0.0f) { bilinear(T, 0, ws, wt, rgb); return; }
    float fl = floorf(lambda);
    int l0 = (int)fl; if (l0 > maxl) l0 = maxl;
    int l1 = l0 + 1 > maxl ? maxl : l0 + 1;
    float fr = l0 == maxl ? 0.0f : lambda - fl;
    float c0[3], c1[3];
    bilinear(T, l0, ws, wt, c0);
    if (l1 == l0 || fr == 0.0f) { rgb[0] = c0[0]; rgb[1] = c0[1]; rgb[2] = c0[2]; return; }
    bilinear(T, l1, ws, wt, c1);
    for (int k = 0; k < 3; k++) rgb[k] = c0[k] * (1.0f - fr) + c1[k] * fr;
}

typedef struct {
    float light_dir[3], amb[3], diff[3], sky[3];
    float box_color[3];
} Light;

static void lit_color(const Light *L, const float *n, const float *C, float *out) {
    float ndl = n[0] * L->light_dir[0] + n[1] * L->light_dir[1] + n[2] * L->light_dir[2];
    if (ndl < 0) ndl = 0;
    for (int k = 0; k < 3; k++) {
        float v = (0.2f * C[k] + L->amb[k] * C[k]) + ndl * L->diff[k] * C[k];
        out[k] = v > 1.0f ? 1.0f : v;
    }
}

/* texture coordinates of the surface `h` where ray (o,d) meets the surface's plane; 0 if behind */
static int surf_texcoord(const RRoom *rr, const Hit *h, const float *o, const float *d, float *s, float *t) {
    const RRoom *r = &rr[h->room];
    float tt;
    if (h->kind == 1 || h->kind == 2) {
        float py = h->kind == 1 ? 0.0f : r->height;
        if (d[1] == 0) return 0;
        tt = (py - o[1]) / d[1];
        if (!(tt > 0)) return 0;
        int tex = h->kind == 1 ? r->floor_tex : r->ceil_tex;
        float sc_s = (float)(512.0 / tex_width(tex)), sc_t = (float)(512.0 / tex_height(tex));
        *s = (o[0] + tt * d[0]) * sc_s; *t = (o[2] + tt * d[2]) * sc_t;
        return 1;
    }
    int sd = h->side;
    float sc_s = (float)(512.0 / tex_width(r->wall_tex)), sc_t = (float)(512.0 / tex_height(r->wall_tex));
    float plane, od, oo; int along;
    if (sd == 0 || sd == 2) { plane = sd == 0 ? r->max_x : r->min_x; od = d[0]; oo = o[0]; along = 2; }
    else { plane = sd == 3 ? r->max_z : r->min_z; od = d[2]; oo = o[2]; along = 0; }
    if (od == 0) return 0;
    tt = (plane - oo) / od;
    if (!(tt > 0)) return 0;
    float hc = o[along] + tt * d[along], y = o[1] + tt * d[1];
    *s = ((hc - r->side[sd].u_org) * r->side[sd].u_sgn) * sc_s; *t = y * sc_t;
    return 1;
}

/* One shade per (pixel, triangle), attributes evaluated at the PIXEL CENTRE (extrapolated beyond the triangle's edges, as a
 * multisampling rasteriser without centroid sampling does): lit vertex colours - fixed-function lighting per vertex with the
 * normal as the modelview matrix leaves it, R n / scale, NOT renormalised (GL_NORMALIZE is off, entity.py:131-141), clamped to 1 -
 * interpolated with the centre ray's barycentrics u / det, v / det; if the centre ray sees the triangle's back or edge (det <= 0)
 * the covering sample's own ray is used.  A textured mesh modulates by its image (trilinear, LOD from the +1 pixel neighbours). */
static void shade_mesh(const RMesh *rm, const Light *L, int tri, const float *dc, const float *dx, const float *dy, const float *ds, float *col) {
    const Mesh *m = rm->m;
    float vc[3][3];
    for (int k = 0; k < 3; k++) {
        const float *n = m->norms + tri * 9 + k * 3;
        float ndl = n[0] * rm->Ll[0] + n[1] * rm->Ll[1] + n[2] * rm->Ll[2];
        if (ndl < 0) ndl = 0;
        for (int q = 0; q < 3; q++) {
            float v = (0.2f * rm->kd[q] + L->amb[q] * rm->kd[q]) + ndl * L->diff[q] * rm->kd[q];
            vc[k][q] = v > 1.0f ? 1.0f : v;
        }
    }
    float ld[3], t, u, v, det, ub, vb;
    mesh_local_dir(rm, dc, ld);
    int centre_ok = mesh_tri(rm, tri, ld, &t, &u, &v, &det, 0);
    if (!centre_ok) { mesh_local_dir(rm, ds, ld); mesh_tri(rm, tri, ld, &t, &u, &v, &det, 0); }
    ub = u / det; vb = v / det;
    for (int q = 0; q < 3; q++) col[q] = fmaf(vb, vc[2][q] - vc[0][q], fmaf(ub, vc[1][q] - vc[0][q], vc[0][q]));
    if (m->tex_id >= 0) {
        const float *tc = m->texcs + tri * 6;
        float s0 = fmaf(vb, tc[4] - tc[0], fmaf(ub, tc[2] - tc[0], tc[0])), t0 = fmaf(vb, tc[5] - tc[1], fmaf(ub, tc[3] - tc[1], tc[1]));
        float s1 = s0, t1 = t0, s2 = s0, t2 = t0;
        int valid = 0;
        if (centre_ok) {
            float l1[3], l2[3], tt, u1, v1, d1, u2, v2, d2;
            mesh_local_dir(rm, dx, l1); mesh_local_dir(rm, dy, l2);
            const int ok1 = mesh_tri(rm, tri, l1, &tt, &u1, &v1, &d1, 0), ok2 = mesh_tri(rm, tri, l2, &tt, &u2, &v2, &d2, 0);
            valid = ok1 && ok2;
            if (valid) {
                u1 /= d1; v1 /= d1; u2 /= d2; v2 /= d2;
                s1 = fmaf(v1, tc[4] - tc[0], fmaf(u1, tc[2] - tc[0], tc[0])); t1 = fmaf(v1, tc[5] - tc[1], fmaf(u1, tc[3] - tc[1], tc[1]));
                s2 = fmaf(v2, tc[4] - tc[0], fmaf(u2, tc[2] - tc[0], tc[0])); t2 = fmaf(v2, tc[5] - tc[1], fmaf(u2, tc[3] - tc[1], tc[1]));
            }
        }
        float texel[3];
        sample_texture(m->tex_id, s0, t0, s1, t1, s2, t2, valid, texel);
        for (int q = 0; q < 3; q++) col[q] = col[q] * (texel[q] * (1.0f / 255.0f));
    }
}

/* ---- ImageFrame / TextFrame (entity.py:148-360): a slab [0, depth] x [-h/2, h/2] x [-w/2, w/2] in the frame's axes whose +x face
 * shows the picture (or one texture per character, a space is plain white), whose +-y and +-z faces are black and whose -x face
 * (against the wall) does not exist; front faces only.  Hit codes: 0 .. n_chars-1 the front's character cell, 100 a black side. */
typedef struct { float pos[3], c, s, sx, hy, hz, cw, lit_front[3]; int n_chars, tex[8]; } RFrame;
static int frame_cell(const RFrame *f, float z) {   /* character index of local z on the front: cell i spans [hz - cw (i + 1), hz - cw i] */
    int i = (int)floorf((f->hz - z) / f->cw);
    return i < 0 ? 0 : (i >= f->n_chars ? f->n_chars - 1 : i);
}
static int trace_frame(const RFrame *f, const float *o, const float *d, float *t_out) {
    float ro[3] = {o[0] - f->pos[0], o[1] - f->pos[1], o[2] - f->pos[2]};
    float lo_[3] = {ro[0] * f->c - ro[2] * f->s, ro[1], ro[0] * f->s + ro[2] * f->c};
    float ld[3] = {fmaf(d[0], f->c, -(d[2] * f->s)), d[1], fmaf(d[0], f->s, d[2] * f->c)};
    float lo[3] = {0.0f, -f->hy, -f->hz}, hi[3] = {f->sx, f->hy, f->hz};
    float tn = -INFINITY, tf = INFINITY; int face = -1;
    for (int a = 0; a < 3; a++) {
        if (ld[a] == 0) { if (lo_[a] < lo[a] || lo_[a] > hi[a]) return -1; continue; }
        float inv = 1.0f / ld[a];
        float t1 = (lo[a] - lo_[a]) * inv, t2 = (hi[a] - lo_[a]) * inv;
        float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
        if (tmin > tn) { tn = tmin; face = a * 2 + (ld[a] > 0 ? 0 : 1); }
        if (tmax < tf) tf = tmax;
    }
    if (face < 0 || !(tn <= tf) || !(tn > 0)) return -1;
    if (face == 0) return -1;   /* entered through the -x face: it is not drawn, and the inner faces are culled */
    *t_out = tn;
    if (face != 1) return 100;
    return frame_cell(f, fmaf(tn, ld[2], lo_[2]));
}
static int frame_front_tc(const RFrame *f, int cell, const float *o, const float *d, float *s, float *t) {
    float ro[3] = {o[0] - f->pos[0], o[1] - f->pos[1], o[2] - f->pos[2]};
    float lo_[3] = {ro[0] * f->c - ro[2] * f->s, ro[1], ro[0] * f->s + ro[2] * f->c};
    float ld[3] = {fmaf(d[0], f->c, -(d[2] * f->s)), d[1], fmaf(d[0], f->s, d[2] * f->c)};
    if (ld[0] == 0) return 0;
    float tt = (f->sx - lo_[0]) / ld[0];
    if (!(tt > 0)) return 0;
    float y = fmaf(tt, ld[1], lo_[1]), z = fmaf(tt, ld[2], lo_[2]);
    float z1 = f->hz - f->cw * (float)cell;          /* the cell's right edge: texcoord 0 (glTexCoord2f(0, *) at z_1) */
    *s = (z1 - z) / f->cw; *t = (y + f->hy) / (2.0f * f->hy);
    return 1;
}
static void shade_frame(const RFrame *f, int code, const float *o, const float *dc, const float *dx, const float *dy, const float *ds, float *col) {
    if (code >= 100) { col[0] = col[1] = col[2] = 0.0f; return; }   /* glColor3f(0, 0, 0): lit black is black */
    const int tex = f->tex[code];
    if (tex < 0) { col[0] = f->lit_front[0]; col[1] = f->lit_front[1]; col[2] = f->lit_front[2]; return; }   /* a space: texturing off */
    float s0 = 0, t0 = 0, s1 = 0, t1 = 0, s2 = 0, t2 = 0;
    int valid = 1;
    if (!frame_front_tc(f, code, o, dc, &s0, &t0)) { frame_front_tc(f, code, o, ds, &s0, &t0); valid = 0; s1 = s2 = s0; t1 = t2 = t0; }
    else valid = frame_front_tc(f, code, o, dx, &s1, &t1) && frame_front_tc(f, code, o, dy, &s2, &t2);
    float texel[3];
    sample_texture(tex, s0, t0, s1, t1, s2, t2, valid, texel);
    for (int q = 0; q < 3; q++) col[q] = f->lit_front[q] * (texel[q] * (1.0f / 255.0f));
}

void mwo_render(MwoEnv *e, int W, int H, uint8_t *rgb, float *depth) {
    const int poly = e->task == MWO_YMAZE;   /* rooms with arbitrary edges: the polygon formulation */
    RRoom *rr = (RRoom *)malloc(sizeof(RRoom) * (size_t)(e->n_rooms > 0 ? 2 * e->n_rooms : 1));   /* per call: thread-safe */
    PRoom *pr = (PRoom *)malloc(sizeof(PRoom) * (size_t)(e->n_rooms > 0 ? e->n_rooms : 1));
    int n_rooms = poly ? build_prooms(e, pr) : build_rrooms(e, rr);
    /* camera: gluPerspective(fov_y, W/H, 0.04, 100), gluLookAt(cam_pos, cam_pos+cam_dir, +Y) */
    double cp[3], cd[3];
    camera(e, cp, cd);
    double fl = sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]);
    double f[3] = {cd[0] / fl, cd[1] / fl, cd[2] / fl};
    double sl = sqrt(f[2] * f[2] + f[0] * f[0]);
    double s[3] = {-f[2] / sl, 0, f[0] / sl};
    double u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    double th = tan(e->cam_fov_y * M_PI / 180 / 2);
    Cam c;
    for (int k = 0; k < 3; k++) { c.eye[k] = (float)cp[k]; c.F[k] = (float)f[k]; c.S[k] = (float)s[k]; c.U[k] = (float)u[k]; }
    c.TH = (float)th; c.TW = (float)(th * ((double)W / (double)H));
    c.W = W; c.H = H; c.invW = 1.0f / (float)W; c.invH = 1.0f / (float)H;
    /* light: GL_POSITION = (light_pos + 1, w = 0) -> directional (miniworld.py:1026) */
    Light L;
    double lp[3] = {e->light_pos[0] + 1, e->light_pos[1] + 1, e->light_pos[2] + 1};
    double ll = sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]);
    for (int k = 0; k < 3; k++) {
        L.light_dir[k] = (float)(lp[k] / ll); L.amb[k] = (float)e->light_ambient[k]; L.diff[k] = (float)e->light_color[k];
        L.sky[k] = (float)e->sky_color[k]; L.box_color[k] = (float)e->box_colors[0][k];
    }
    const float white[3] = {1, 1, 1};
    float lit_floor[3], lit_ceil[3], lit_wall[4][3];
    { float n[3] = {0, 1, 0}; lit_color(&L, n, white, lit_floor); }
    { float n[3] = {0, -1, 0}; lit_color(&L, n, white, lit_ceil); }
    { static const float wn[4][3] = {{-1, 0, 0}, {0, 0, 1}, {1, 0, 0}, {0, 0, -1}};
      for (int k = 0; k < 4; k++) lit_color(&L, wn[k], white, lit_wall[k]); }
    RBox bxs[MWO_MAX_BOXES];
    RMesh rms[MWO_MAX_BOXES];
    RFrame rfs[MWO_MAX_BOXES];
    int ekind[MWO_MAX_BOXES];   /* what to draw for slot b: -1 nothing (left the list), else MWO_ENT_* */
    float lit_boxes[MWO_MAX_BOXES][6][3];
    for (int b = 0; b < e->n_boxes; b++) {
        Ent frame_ent = e->boxes[b];
        if (e->render_step_frame) {   /* the frame the reference's step() renders: BEFORE the task rule removes / respawns an entity */
            memcpy(frame_ent.pos, e->frame_pos[b], sizeof(frame_ent.pos)); frame_ent.dir = e->frame_dir[b]; frame_ent.alive = e->frame_alive[b];
        }
        const Ent *be = &frame_ent;
        ekind[b] = be->alive ? be->kind : -1;
        if (ekind[b] == MWO_ENT_MESH) {
            RMesh *rm = &rms[b];
            rm->m = &g_mesh[be->geom];
            if (!rm->m->n_tris) fail("mesh not loaded (mwo_set_mesh)");
            for (int k = 0; k < 3; k++) { rm->pos[k] = (float)be->pos[k]; rm->kd[k] = (float)e->box_colors[b][k]; }
            rm->c = (float)cos(be->dir); rm->s = (float)sin(be->dir); rm->inv_s = (float)(1.0 / be->scale);
            const float ro[3] = {c.eye[0] - rm->pos[0], c.eye[1] - rm->pos[1], c.eye[2] - rm->pos[2]};
            rm->lo[0] = (ro[0] * rm->c - ro[2] * rm->s) * rm->inv_s; rm->lo[1] = ro[1] * rm->inv_s; rm->lo[2] = (ro[0] * rm->s + ro[2] * rm->c) * rm->inv_s;
            rm->Ll[0] = (L.light_dir[0] * rm->c - L.light_dir[2] * rm->s) * rm->inv_s; rm->Ll[1] = L.light_dir[1] * rm->inv_s;
            rm->Ll[2] = (L.light_dir[0] * rm->s + L.light_dir[2] * rm->c) * rm->inv_s;
            double lc[3], ext2 = 0;   /* bounding sphere of the mesh's box, 5 % larger */
            for (int k = 0; k < 3; k++) { lc[k] = 0.5 * ((double)rm->m->min_c[k] + rm->m->max_c[k]); double h = 0.5 * ((double)rm->m->max_c[k] - rm->m->min_c[k]); ext2 += h * h; }
            const double cd_ = cos(be->dir), sd_ = sin(be->dir);
            rm->gate_c[0] = be->pos[0] + be->scale * (lc[0] * cd_ + lc[2] * sd_);
            rm->gate_c[1] = be->pos[1] + be->scale * lc[1];
            rm->gate_c[2] = be->pos[2] + be->scale * (-lc[0] * sd_ + lc[2] * cd_);
            rm->gate_r2 = ext2 * be->scale * be->scale * 1.05 * 1.05;
            continue;
        }
        if (ekind[b] == MWO_ENT_IMAGE || ekind[b] == MWO_ENT_TEXT) {
            RFrame *rf = &rfs[b];
            for (int k = 0; k < 3; k++) rf->pos[k] = (float)be->pos[k];
            rf->c = (float)cos(be->dir); rf->s = (float)sin(be->dir);
            rf->sx = (float)be->frame_d; rf->hy = (float)(be->frame_h / 2); rf->hz = (float)(be->frame_w / 2);
            rf->n_chars = ekind[b] == MWO_ENT_TEXT ? be->n_chars : 1;
            rf->cw = ekind[b] == MWO_ENT_TEXT ? (float)be->frame_h : (float)be->frame_w;   /* char_width = self.height (entity.py:303) */
            for (int k = 0; k < 8; k++) rf->tex[k] = be->tex[k];
            float n[3] = {rf->c, 0.0f, -rf->s};   /* the front's normal (1, 0, 0) turned by the heading */
            lit_color(&L, n, white, rf->lit_front);
            continue;
        }
        if (ekind[b] != MWO_ENT_BOX) continue;
        RBox bx;
        float bcol[3];
        for (int k = 0; k < 3; k++) { bx.pos[k] = (float)be->pos[k]; bcol[k] = (float)e->box_colors[b][k]; }
        bx.c = (float)cos(be->dir); bx.s = (float)sin(be->dir);
        bx.half[0] = (float)(e->box_s[b] / 2); bx.half[2] = (float)(e->box_s[b] / 2); bx.half[1] = 0; bx.sy = (float)e->box_s[b];
        /* world normal of local normal n: R_y(dir) n = (nx c + nz s, ny, -nx s + nz c) */
        static const float ln[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        for (int k = 0; k < 6; k++) {
            float n[3] = {ln[k][0] * bx.c + ln[k][2] * bx.s, ln[k][1], -ln[k][0] * bx.s + ln[k][2] * bx.c};
            lit_color(&L, n, bcol, lit_boxes[b][k]);
        }
        bxs[b] = bx;
    }
    /* room containing the eye */
    int cam_room = -1;
    for (int i = 0; i < n_rooms && cam_room < 0; i++) {
        if (!poly) { if (c.eye[0] >= rr[i].min_x && c.eye[0] <= rr[i].max_x && c.eye[2] >= rr[i].min_z && c.eye[2] <= rr[i].max_z) cam_room = i; }
        else {   /* first room, in creation order, whose every edge has the eye on its inner side (bounds inclusive) */
            int in = !pr[i].culled;
            for (int k = 0; k < pr[i].ne; k++) {
                const PEdge *pe = &pr[i].e[k];
                if (!(fmaf(pe->nz, c.eye[2] - pe->pz, pe->nx * (c.eye[0] - pe->px)) >= 0.0f)) in = 0;
            }
            if (in) cam_room = i;
        }
    }
    const float n_ = 0.04f, f_ = 100.0f;
    const float zA = (f_ + n_) / (f_ - n_), zB = (2.0f * f_ * n_) / (f_ - n_);
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
            Hit hits[8];
            for (int k = 0; k < 8; k++) {
                float d[3];
                make_ray(&c, cx + SAMPLE_X[k] * 0.0625f, cy + SAMPLE_Y[k] * 0.0625f, d);
                hits[k] = poly ? trace_rooms_poly(pr, n_rooms, cam_room, c.eye, d) : trace_rooms(rr, n_rooms, cam_room, c.eye, d);
                for (int b = 0; b < e->n_boxes; b++) { /* slot order; a later entity wins only when strictly nearer */
                    float tb;
                    if (ekind[b] == MWO_ENT_BOX) {
                        int fc = trace_box(&bxs[b], c.eye, d, &tb);
                        if (fc >= 0 && tb < hits[k].t) { hits[k].kind = 4; hits[k].t = tb; hits[k].room = b; hits[k].side = fc; }
                    } else if (ekind[b] == MWO_ENT_MESH) {
                        int tri = trace_mesh(&rms[b], c.eye, d, &tb);
                        if (tri >= 0 && tb < hits[k].t) { hits[k].kind = 5; hits[k].t = tb; hits[k].room = b; hits[k].side = tri; }
                    } else if (ekind[b] == MWO_ENT_IMAGE || ekind[b] == MWO_ENT_TEXT) {
                        int fc = trace_frame(&rfs[b], c.eye, d, &tb);
                        if (fc >= 0 && tb < hits[k].t) { hits[k].kind = 6; hits[k].t = tb; hits[k].room = b; hits[k].side = fc; }
                    }
                }
            }
            float dc[3], dx[3], dy[3];
            make_ray(&c, cx, cy, dc); make_ray(&c, cx + 1.0f, cy, dx); make_ray(&c, cx, cy + 1.0f, dy);
            float acc[3] = {0, 0, 0};
            int done_mask = 0;
            for (int k = 0; k < 8; k++) {
                if (done_mask & (1 << k)) continue;
                int cnt = 0;
                for (int j = k; j < 8; j++)
                    if (!(done_mask & (1 << j)) && hits[j].kind == hits[k].kind && hits[j].room == hits[k].room && hits[j].side == hits[k].side) { cnt++; done_mask |= 1 << j; }
                float col[3];
                const Hit *h = &hits[k];
                if (h->kind == 0) { col[0] = L.sky[0]; col[1] = L.sky[1]; col[2] = L.sky[2]; }
                else if (h->kind == 4) { const float *lb = lit_boxes[h->room][h->side]; col[0] = lb[0]; col[1] = lb[1]; col[2] = lb[2]; }
                else if (h->kind == 5) {
                    float ds[3]; make_ray(&c, cx + SAMPLE_X[k] * 0.0625f, cy + SAMPLE_Y[k] * 0.0625f, ds);
                    shade_mesh(&rms[h->room], &L, h->side, dc, dx, dy, ds, col);
                }
                else if (h->kind == 6) {
                    float ds[3]; make_ray(&c, cx + SAMPLE_X[k] * 0.0625f, cy + SAMPLE_Y[k] * 0.0625f, ds);
                    shade_frame(&rfs[h->room], h->side, c.eye, dc, dx, dy, ds, col);
                }
                else {
                    float s0 = 0, t0 = 0, s1 = 0, t1 = 0, s2 = 0, t2 = 0;
                    int tex;
                    const float *lit;
                    float lit_edge[3];
                    if (!poly) {
                        tex = h->kind == 1 ? rr[h->room].floor_tex : h->kind == 2 ? rr[h->room].ceil_tex : rr[h->room].wall_tex;
                        lit = h->kind == 1 ? lit_floor : h->kind == 2 ? lit_ceil : lit_wall[h->side];
                    } else {
                        tex = h->kind == 1 ? pr[h->room].floor_tex : h->kind == 2 ? pr[h->room].ceil_tex : pr[h->room].wall_tex;
                        lit = h->kind == 1 ? lit_floor : lit_ceil;
                        if (h->kind == 3) {   /* flat face with the edge's inward normal */
                            float n[3] = {pr[h->room].e[h->side].nx, 0.0f, pr[h->room].e[h->side].nz};
                            lit_color(&L, n, white, lit_edge);
                            lit = lit_edge;
                        }
                    }
#define SURF_TC(ray, ps, pt) (poly ? surf_texcoord_poly(pr, h, c.eye, ray, ps, pt) : surf_texcoord(rr, h, c.eye, ray, ps, pt))
                    int okc = SURF_TC(dc, &s0, &t0);
                    int valid = 1;
                    if (!okc) { /* centre ray misses the plane: shade at the sample's own hit point */
                        float d[3]; make_ray(&c, cx + SAMPLE_X[k] * 0.0625f, cy + SAMPLE_Y[k] * 0.0625f, d);
                        SURF_TC(d, &s0, &t0); valid = 0; s1 = s2 = s0; t1 = t2 = t0;
                    } else {
                        valid = SURF_TC(dx, &s1, &t1) && SURF_TC(dy, &s2, &t2);
                    }
                    float texel[3];
                    sample_texture(tex, s0, t0, s1, t1, s2, t2, valid, texel);
                    for (int q = 0; q < 3; q++) col[q] = lit[q] * (texel[q] * (1.0f / 255.0f));
                }
                for (int q = 0; q < 3; q++) acc[q] += (float)cnt * col[q];
            }
            for (int q = 0; q < 3; q++) {
                float v = acc[q] * 0.125f;
                v = v < 0 ? 0 : (v > 1 ? 1 : v);
                rgb[(py * W + px) * 3 + q] = (uint8_t)(int)floorf(v * 255.0f + 0.5f);
            }
            if (depth) {
                int z16 = 65535;
                if (hits[0].kind != 0) {
                    float zn = zA - zB / hits[0].t;
                    float dd = 0.5f * zn + 0.5f;
                    int z = (int)floorf(dd * 65535.0f + 0.5f);
                    z16 = z < 0 ? 0 : (z > 65535 ? 65535 : z);
                }
                /* get_depth_map, opengl.py:362-367, float32 arithmetic */
                float dm = (float)z16 / 65535.0f;
                float clip_z = (dm - 0.5f) * 2.0f;
                float wz = (float)(-2.0 * 100.0 * 0.04) / (clip_z * (float)(100.0 - 0.04) - (float)(100.0 + 0.04));
                depth[py * W + px] = wz;
            }
        }
    free(rr);
    free(pr);
}

/* ---- get_visible_ents (miniworld.py:1222-1315): the rooms are drawn into the observation frame buffer (8 samples, the
 * camera of render_obs), then for every entity but the agent, in list order, an axis-aligned 0.2 m cube at its position
 * (x, z +- 0.1, y .. y + 0.2; drawBox: no rotation) inside a GL_ANY_SAMPLES_PASSED query.  Depth test GL_LESS with depth
 * writes on: a cube is visible iff at some sample its front face is nearer than the room surface and than every cube drawn
 * before it.  Depths are compared as ray parameters of the same ray (monotonic in the depth-buffer value; the 24-bit
 * quantisation and the near plane are not modelled).  The reference never calls this function and no output of it exists
 * anywhere: PARITY UNPINNED - this restatement is the spec the HIP kernel is tested against. */
static int cube_entry(const float *lo, const float *hi, const float *o, const float *d, float *t_out) {
    float tn = -INFINITY, tf = INFINITY;
    for (int a = 0; a < 3; a++) {
        if (d[a] == 0) { if (o[a] < lo[a] || o[a] > hi[a]) return 0; continue; }
        float inv = 1.0f / d[a];
        float t1 = (lo[a] - o[a]) * inv, t2 = (hi[a] - o[a]) * inv;
        float tmin = t1 < t2 ? t1 : t2, tmax = t1 < t2 ? t2 : t1;
        if (tmin > tn) tn = tmin;
        if (tmax < tf) tf = tmax;
    }
    if (!(tn <= tf) || !(tn > 0)) return 0;   /* eye inside the cube: only back faces, culled */
    *t_out = tn;
    return 1;
}

uint32_t mwo_visible_ents(MwoEnv *e, int W, int H) {
    const int poly = e->task == MWO_YMAZE;
    RRoom *rr = (RRoom *)malloc(sizeof(RRoom) * (size_t)(e->n_rooms > 0 ? 2 * e->n_rooms : 1));
    PRoom *pr = (PRoom *)malloc(sizeof(PRoom) * (size_t)(e->n_rooms > 0 ? e->n_rooms : 1));
    int n_rooms = poly ? build_prooms(e, pr) : build_rrooms(e, rr);
    double cp[3], cd[3];
    camera(e, cp, cd);
    double fl = sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]);
    double f[3] = {cd[0] / fl, cd[1] / fl, cd[2] / fl};
    double sl = sqrt(f[2] * f[2] + f[0] * f[0]);
    double s[3] = {-f[2] / sl, 0, f[0] / sl};
    double u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    double th = tan(e->cam_fov_y * M_PI / 180 / 2);
    Cam c;
    for (int k = 0; k < 3; k++) { c.eye[k] = (float)cp[k]; c.F[k] = (float)f[k]; c.S[k] = (float)s[k]; c.U[k] = (float)u[k]; }
    c.TH = (float)th; c.TW = (float)(th * ((double)W / (double)H));
    c.W = W; c.H = H; c.invW = 1.0f / (float)W; c.invH = 1.0f / (float)H;
    int cam_room = -1;
    for (int i = 0; i < n_rooms && cam_room < 0; i++) {
        if (!poly) { if (c.eye[0] >= rr[i].min_x && c.eye[0] <= rr[i].max_x && c.eye[2] >= rr[i].min_z && c.eye[2] <= rr[i].max_z) cam_room = i; }
        else {
            int in = !pr[i].culled;
            for (int k = 0; k < pr[i].ne; k++) {
                const PEdge *pe = &pr[i].e[k];
                if (!(fmaf(pe->nz, c.eye[2] - pe->pz, pe->nx * (c.eye[0] - pe->px)) >= 0.0f)) in = 0;
            }
            if (in) cam_room = i;
        }
    }
    /* the cubes in drawing order: self.entities without the agent (e->order: entities that were removed are not in it) */
    float lo[MWO_MAX_BOXES][3], hi[MWO_MAX_BOXES][3];
    int slot[MWO_MAX_BOXES], n_cubes = 0;
    if (e->task >= MWO_PICKUPOBJS) {
        for (int q = 0; q < e->n_order; q++)
            if (e->order[q] != AGENT_SLOT) slot[n_cubes++] = e->order[q];
    } else {   /* the box tasks place their boxes in slot order */
        for (int b = 0; b < e->n_boxes; b++) slot[n_cubes++] = b;
    }
    for (int b = 0; b < n_cubes; b++) {   /* glVertex3f arguments: float32 of the float64 sums */
        const double *p = e->boxes[slot[b]].pos;
        lo[b][0] = (float)(p[0] - 0.1); hi[b][0] = (float)(p[0] + 0.1);
        lo[b][1] = (float)p[1];         hi[b][1] = (float)(p[1] + 0.2);
        lo[b][2] = (float)(p[2] - 0.1); hi[b][2] = (float)(p[2] + 0.1);
    }
    uint32_t mask = 0;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
            for (int k = 0; k < 8; k++) {
                float d[3], tb[MWO_MAX_BOXES];
                int any = 0;
                make_ray(&c, cx + SAMPLE_X[k] * 0.0625f, cy + SAMPLE_Y[k] * 0.0625f, d);
                for (int b = 0; b < n_cubes; b++) { tb[b] = INFINITY; if (cube_entry(lo[b], hi[b], c.eye, d, &tb[b])) any = 1; else tb[b] = INFINITY; }
                if (!any) continue;
                Hit h = poly ? trace_rooms_poly(pr, n_rooms, cam_room, c.eye, d) : trace_rooms(rr, n_rooms, cam_room, c.eye, d);
                float depth = h.kind != 0 ? h.t : INFINITY;   /* the depth buffer at this sample so far */
                for (int b = 0; b < n_cubes; b++)
                    if (tb[b] < depth) { mask |= 1u << slot[b]; depth = tb[b]; }
            }
        }
    free(rr);
    free(pr);
    return mask;
}

/* ---- render_top_view (miniworld.py:1087-1158): glOrtho over the floorplan's extents + 1 m (widened to the frame's aspect),
 * looking straight down (modelview x -> x, z -> -y, y -> z), _render_world(render_agent=True).  Walls and box sides are
 * vertical: edge-on, they cover nothing.  Ceilings face away from the viewer (back-face culled) - except those of rooms whose
 * outline runs the other way round (YMaze's connector slivers), which face UP.  What a sample sees is the highest of: such a
 * ceiling, the agent's triangle at agent.height (entity.py:494-514; it is lit with the normal the last box face left current,
 * (0, -1, 0): ambient terms only), a box's top face, the floor of the first room (creation order) that holds the point, else the
 * clear colour; equal heights go to the surface drawn first (GL_LESS): rooms, boxes in list order, agent.
 * Frozen choices (float32): x = fmaf(wx, XS, X0), z = fmaf(-wy, ZS, Z1) for window coordinates (wx, wy up), XS = width / W,
 * ZS = height / H from the float64 extents; containment tests inclusive; 8 coverage samples and one shade per (pixel, surface)
 * at the pixel centre as in mwo_render; LOD from the +1 pixel neighbours (x + XS, z) and (x, z - ZS). */
typedef struct { int kind, idx; float y; } TopHit;   /* kind: 0 clear colour, 1 floor, 2 up-facing ceiling, 4 box top, 5 agent */

void mwo_render_top(MwoEnv *e, int W, int H, uint8_t *rgb) {
    const int poly = e->task == MWO_YMAZE;
    RRoom *rr = (RRoom *)malloc(sizeof(RRoom) * (size_t)(e->n_rooms > 0 ? 2 * e->n_rooms : 1));
    PRoom *pr = (PRoom *)malloc(sizeof(PRoom) * (size_t)(e->n_rooms > 0 ? e->n_rooms : 1));
    int n_rooms = poly ? build_prooms(e, pr) : build_rrooms(e, rr);
    /* extents, miniworld.py:576-579 and 1108-1131 */
    double min_x = e->rooms[0].min_x, max_x = e->rooms[0].max_x, min_z = e->rooms[0].min_z, max_z = e->rooms[0].max_z;
    for (int i = 1; i < e->n_rooms; i++) {
        if (e->rooms[i].min_x < min_x) min_x = e->rooms[i].min_x;
        if (e->rooms[i].max_x > max_x) max_x = e->rooms[i].max_x;
        if (e->rooms[i].min_z < min_z) min_z = e->rooms[i].min_z;
        if (e->rooms[i].max_z > max_z) max_z = e->rooms[i].max_z;
    }
    min_x -= 1; max_x += 1; min_z -= 1; max_z += 1;
    {
        double width = max_x - min_x, height = max_z - min_z, aspect = width / height, fb_aspect = (double)W / (double)H;
        if (aspect > fb_aspect) { double new_h = width / fb_aspect, h_diff = new_h - height; min_z -= h_diff / 2; max_z += h_diff / 2; }
        else if (aspect < fb_aspect) { double new_w = height * fb_aspect, w_diff = new_w - width; min_x -= w_diff / 2; max_x += w_diff / 2; }
    }
    const float X0 = (float)min_x, XS = (float)((max_x - min_x) / (double)W), Z1 = (float)max_z, ZS = (float)((max_z - min_z) / (double)H);
    Light L;
    double lp[3] = {e->light_pos[0] + 1, e->light_pos[1] + 1, e->light_pos[2] + 1};
    double ll = sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]);
    for (int k = 0; k < 3; k++) {
        L.light_dir[k] = (float)(lp[k] / ll); L.amb[k] = (float)e->light_ambient[k]; L.diff[k] = (float)e->light_color[k];
        L.sky[k] = (float)e->sky_color[k];
    }
    const float white[3] = {1, 1, 1}, red[3] = {1, 0, 0}, up[3] = {0, 1, 0}, down[3] = {0, -1, 0};
    float lit_floor[3], lit_ceil[3], lit_agent[3], lit_top[MWO_MAX_BOXES][3];
    lit_color(&L, up, white, lit_floor); lit_color(&L, down, white, lit_ceil); lit_color(&L, down, red, lit_agent);
    float bx[MWO_MAX_BOXES], bz[MWO_MAX_BOXES], bc[MWO_MAX_BOXES], bs[MWO_MAX_BOXES], bh[MWO_MAX_BOXES], btop[MWO_MAX_BOXES];
    for (int b = 0; b < e->n_boxes; b++) {
        const Ent *be = &e->boxes[b];
        float col[3];
        for (int k = 0; k < 3; k++) col[k] = (float)e->box_colors[b][k];
        lit_color(&L, up, col, lit_top[b]);
        bx[b] = (float)be->pos[0]; bz[b] = (float)be->pos[2]; bc[b] = (float)cos(be->dir); bs[b] = (float)sin(be->dir);
        bh[b] = (float)(e->box_s[b] / 2); btop[b] = (float)be->pos[1] + (float)e->box_s[b];
    }
    /* the agent's triangle, entity.py:502-510 */
    float tri[3][2];
    {
        const Ent *a = &e->agent;
        double dvx = cos(a->dir) * a->radius, dvz = -sin(a->dir) * a->radius, rvx = sin(a->dir) * a->radius, rvz = cos(a->dir) * a->radius;
        tri[0][0] = (float)(a->pos[0] + dvx); tri[0][1] = (float)(a->pos[2] + dvz);
        tri[1][0] = (float)(a->pos[0] + 0.75 * (rvx - dvx)); tri[1][1] = (float)(a->pos[2] + 0.75 * (rvz - dvz));
        tri[2][0] = (float)(a->pos[0] + 0.75 * (-rvx - dvx)); tri[2][1] = (float)(a->pos[2] + 0.75 * (-rvz - dvz));
    }
    const float agent_y = (float)(e->agent.pos[1] + e->agent.height);
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const float cx = (float)px + 0.5f, cy = (float)(H - 1 - py) + 0.5f;
            TopHit hits[8];
            for (int k = 0; k < 8; k++) {
                const float wx = cx + SAMPLE_X[k] * 0.0625f, wy = cy + SAMPLE_Y[k] * 0.0625f;
                const float x = fmaf(wx, XS, X0), z = fmaf(-wy, ZS, Z1);
                TopHit h = {0, 0, -INFINITY};
                for (int i = 0; i < n_rooms; i++) {   /* rooms first (drawn first): floor at 0, an up-facing ceiling at its height */
                    int in, culled = 0; float height;
                    if (!poly) { in = x >= rr[i].min_x && x <= rr[i].max_x && z >= rr[i].min_z && z <= rr[i].max_z; height = rr[i].height; }
                    else {
                        culled = pr[i].culled; height = pr[i].height; in = 1;
                        for (int q = 0; q < pr[i].ne; q++) {
                            const PEdge *pe = &pr[i].e[q];
                            float side = fmaf(pe->nz, z - pe->pz, pe->nx * (x - pe->px));
                            if (culled ? !(side <= 0.0f) : !(side >= 0.0f)) in = 0;   /* a reversed outline's normals point outwards */
                        }
                    }
                    if (!in) continue;
                    if (!culled && 0.0f > h.y) { h.kind = 1; h.idx = i; h.y = 0.0f; }
                    if (culled && !pr[i].no_ceiling && height > h.y) { h.kind = 2; h.idx = i; h.y = height; }
                }
                for (int b = 0; b < e->n_boxes; b++) {
                    const float rx = x - bx[b], rz = z - bz[b];
                    const float lx = rx * bc[b] - rz * bs[b], lz = rx * bs[b] + rz * bc[b];
                    if (fabsf(lx) <= bh[b] && fabsf(lz) <= bh[b] && btop[b] > h.y) { h.kind = 4; h.idx = b; h.y = btop[b]; }
                }
                {
                    float s0 = 0, s1 = 0, s2 = 0;
                    for (int q = 0; q < 3; q++) {
                        const float *a = tri[q], *b = tri[(q + 1) % 3];
                        float ef = fmaf(b[0] - a[0], z - a[1], -((b[1] - a[1]) * (x - a[0])));
                        if (q == 0) s0 = ef; else if (q == 1) s1 = ef; else s2 = ef;
                    }
                    int in = (s0 >= 0 && s1 >= 0 && s2 >= 0) || (s0 <= 0 && s1 <= 0 && s2 <= 0);
                    if (in && agent_y > h.y) { h.kind = 5; h.idx = 0; h.y = agent_y; }
                }
                hits[k] = h;
            }
            const float xc = fmaf(cx, XS, X0), zc = fmaf(-cy, ZS, Z1);
            float acc[3] = {0, 0, 0};
            int done_mask = 0;
            for (int k = 0; k < 8; k++) {
                if (done_mask & (1 << k)) continue;
                int cnt = 0;
                for (int j = k; j < 8; j++)
                    if (!(done_mask & (1 << j)) && hits[j].kind == hits[k].kind && hits[j].idx == hits[k].idx) { cnt++; done_mask |= 1 << j; }
                float col[3];
                const TopHit *h = &hits[k];
                if (h->kind == 0) { col[0] = L.sky[0]; col[1] = L.sky[1]; col[2] = L.sky[2]; }
                else if (h->kind == 4) { col[0] = lit_top[h->idx][0]; col[1] = lit_top[h->idx][1]; col[2] = lit_top[h->idx][2]; }
                else if (h->kind == 5) { col[0] = lit_agent[0]; col[1] = lit_agent[1]; col[2] = lit_agent[2]; }
                else {
                    int tex = h->kind == 1 ? (poly ? pr[h->idx].floor_tex : rr[h->idx].floor_tex) : pr[h->idx].ceil_tex;
                    const float *lit = h->kind == 1 ? lit_floor : lit_ceil;
                    float sc_s = (float)(512.0 / tex_width(tex)), sc_t = (float)(512.0 / tex_height(tex));
                    float texel[3];
                    sample_texture(tex, xc * sc_s, zc * sc_t, (xc + XS) * sc_s, zc * sc_t, xc * sc_s, (zc - ZS) * sc_t, 1, texel);
                    for (int q = 0; q < 3; q++) col[q] = lit[q] * (texel[q] * (1.0f / 255.0f));
                }
                for (int q = 0; q < 3; q++) acc[q] += (float)cnt * col[q];
            }
            for (int q = 0; q < 3; q++) {
                float v = acc[q] * 0.125f;
                v = v < 0 ? 0 : (v > 1 ? 1 : v);
                rgb[(py * W + px) * 3 + q] = (uint8_t)(int)floorf(v * 255.0f + 0.5f);
            }
        }
    free(rr);
    free(pr);
}

/* ====================================================================== bench helper */
static uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

double mwo_bench_loop(MwoEnv *e, int n_steps, uint64_t action_seed, uint64_t env_index, int W, int H, int want_depth, int constant_action, int n_actions) {
    if (n_actions <= 0) n_actions = 3;
    uint8_t *rgb = malloc((size_t)W * H * 3);
    float *dep = want_depth ? malloc(sizeof(float) * W * H) : NULL;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_steps; t++) {
        int a = constant_action >= 0 ? constant_action
                                     : (int)((splitmix(action_seed ^ splitmix((uint64_t)t * 0x100000001B3ull + env_index)) >> 33) % (uint64_t)n_actions);
        double r; int d;
        mwo_step(e, a, &r, &d);
        if (d) mwo_reset(e);
        e->render_step_frame = !d;   /* the step's own frame (an object just picked up is still drawn), as the GPU path renders it */
        mwo_render(e, W, H, rgb, dep);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(rgb); free(dep);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
