/* miniworld_batch.h - C ABI of the MI355X-native batched MiniWorld stepper + renderer.
 *
 * The reference (mjsargent/gym-miniworld) is pure Python and has no FFI for this path; its
 * boundary is two Python protocols.  Each entry point below names the reference
 * interface it replaces (file:line under the reference root); the Python binding a maintainer
 * would add is shown in INTEGRATION.md and implemented in gym_miniworld_amd/batch.py.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * MWB_E* code, with a message retrievable through mwb_last_error(); no exception crosses the
 * ABI.  One handle owns the state of `num_envs` environments on one GPU; a handle is not
 * thread-safe.  Device work is enqueued on the caller's HIP stream (void* = hipStream_t) and is
 * asynchronous unless stated; outputs are library-owned device buffers that stay valid and are
 * overwritten by the next mwb_step / mwb_reset / mwb_render on the same handle.
 */
#ifndef MINIWORLD_BATCH_H
#define MINIWORLD_BATCH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MWB_ABI_VERSION 5

enum { MWB_OK = 0, MWB_EINVAL = -1, MWB_EHIP = -2, MWB_ENOMEM = -3, MWB_ESTATE = -4 };

/* tasks: the reference env classes registered as MiniWorld-<Class>-v0 (envs/__init__.py:43-49) */
enum { MWB_TASK_HALLWAY = 0,   /* envs/hallway.py   task_args = {length}                    */
       MWB_TASK_ONEROOM = 1,   /* envs/oneroom.py   task_args = {size}                      */
       MWB_TASK_FOURROOMS = 2, /* envs/fourrooms.py task_args = {}                          */
       MWB_TASK_MAZE = 3,      /* envs/maze.py      task_args = {num_rows,num_cols,room_size} */
       /* the T-maze family, envs/tmaze.py:
        * TMaze (7-74: goal_pos None -> random arm), TMazeLeft / TMazeRight (69-75: goal_pos given),
        * TMazeDynamic (77-106: the goal alternates between (10,0,-6) and (10,0,6) every sub_task_length-th
        * reset).  task_args = {goal_pos given (0/1), goal x, goal z, sub_task_length (> 0: TMazeDynamic)} */
       MWB_TASK_TMAZE = 4,
       /* TMazeTwoBoxDynamic (108-217) and its *Features* copies (220-650): a red box at (10,0,-6), a blue box
        * at (10,0,6); reaching the goal box ends the episode with +_reward(), the other with -_reward(), and
        * goal / penalty swap by the class' rule.  task_args = {rule, -, -, sub_task_length}: rule 0 = every
        * sub_task_length-th reset (TMazeTwoBoxDynamic.reset 210-217); rule 1 = at every reset once more than
        * sub_task_length steps have been taken - the counter is never cleared, as in the reference - and
        * info['feature'] = [near(blue), near(red)] is produced (*Features*.step 299-320, reset 322-330) */
       MWB_TASK_TMAZE_TWOBOX = 5,
       /* the sim-to-real rinks, envs/simtorealgoto.py and envs/simtorealpush.py: one square room of random size
        * without ceiling (low walls, sky above), random wall / floor texture families, a robot of radius 0.11
        * and per-episode box sizes; pass their `sim_params` table (params.py sim_to_real_params) and
        * domain_rand = 1 (the classes force it).  GoTo: reach the red box (Discrete(3)).  Push: drive the red box
        * to the yellow one - a forward move first shoves any box it would touch (simtorealpush.py:109-125, with an
        * RNG draw for the box's new heading), move_back is allowed (Discrete(4)), reward 1 when the boxes are
        * within goal_dist = 1.5 (size1 + size2).  task_args = {} */
       MWB_TASK_SIM2REAL_GOTO = 6,
       MWB_TASK_SIM2REAL_PUSH = 7,
       /* envs/putnext.py: one size x size room with six boxes - COLOR_NAMES order (entity.py:18): blue, green,
        * grey, purple, red, yellow - of sizes drawn per episode; the full MiniWorldEnv action set Discrete(8) incl. pickup (4)
        * and drop (5) with the carry physics of miniworld.py:594-606,622-631,645-654,682-702 (toggle 6 and done 7 do
        * nothing); reward + done once the red box is next to the yellow one and nothing is carried.  task_args = {size} */
       MWB_TASK_PUTNEXT = 8,
       /* envs/ymaze.py:8-103 (YMaze, YMazeLeft, YMazeRight): a corridor, a triangular hub and two arms rotated by -+120
        * degrees - rooms are convex polygons with 3 or 4 arbitrary edges (the polygon room table below); reward + done
        * near the box, info['goal_pos'] = box position; Discrete(3).  task_args = {goal given (0 = random arm), goal x, goal z} */
       MWB_TASK_YMAZE = 9,
       /* ---- tasks with a general entity list: mesh entities (MeshEnt / Key / Ball, entity.py:100-146,410-434),
        * image and text frames (entity.py:148-360), entities that leave the list or re-enter it at its end.  Meshes must be
        * registered with mwb_set_mesh / mwb_set_mesh_dims before the first reset. */
       /* envs/pickupobjs.py: num_objs objects (Ball 0.9, Box 0.9 or Key of a random colour each) in a size x size yard without
        * ceiling; Discrete(5); picking an object up removes it (after the step's frame was rendered) for reward 1; done when
        * all are gone.  task_args = {size, num_objs} */
       MWB_TASK_PICKUPOBJS = 10,
       /* envs/roomobjs.py: a box, a ball and a key of random colours, agent radius 1.5, no reward, no time limit
        * (max_episode_steps = INT32_MAX); the base class' Discrete(8).  task_args = {size} */
       MWB_TASK_ROOMOBJS = 11,
       /* envs/collecthealth.py: 18 medkits; health -= 2 per step, a pickup respawns the kit at a random place (it moves to the
        * END of the entity list) and restores health to 100; reward 2 per step alive, -100 and done at health <= 0;
        * info['health'] is delivered in mwb_outputs.feature[0].  task_args = {size} */
       MWB_TASK_COLLECTHEALTH = 12,
       /* envs/threerooms.py: three rooms, two boxes, the Mila logo as an ImageFrame, a duckie, a key, a ball; no reward */
       MWB_TASK_THREEROOMS = 13,
       /* envs/sign.py: three rooms in a U, (blue, red, green) x (box, big key) at fixed places, a TextFrame naming a colour;
        * Discrete(4) where action 3 moves back AND ends the episode; touching an object ends it with +1 (the named colour and the
        * goal's shape) or -1.  task_args = {size, color_index, goal}; pass Sign's parameter table (forward_step 0.7, turn_step 45) */
       MWB_TASK_SIGN = 14,
       /* envs/sidewalk.py: a sidewalk beside a long street (no ceilings), a far building and five cones (static meshes), a red
        * box; stepping into the street ends the episode with reward 0 */
       MWB_TASK_SIDEWALK = 15,
       /* envs/wallgap.py: two yards (no ceilings) joined by a gap in a wall, a red box, the far building */
       MWB_TASK_WALLGAP = 16,
       MWB_NUM_TASKS };

/* entity kinds and mesh geometries of the general entity list */
enum { MWB_ENT_BOX = 0, MWB_ENT_MESH = 1, MWB_ENT_IMAGE = 2, MWB_ENT_TEXT = 3 };
enum { MWB_MESH_BALL = 0, MWB_MESH_KEY, MWB_MESH_MEDKIT, MWB_MESH_DUCKIE, MWB_MESH_BUILDING, MWB_MESH_CONE, MWB_NUM_MESHES };
/* mwb_state.ent_meta word: kind | geometry << 4 | static << 8 | alive << 9 | radius is a float32 scalar << 10 | (colour index + 1) << 12 */
#define MWB_META_KIND(m) ((m) & 15)
#define MWB_META_GEOM(m) (((m) >> 4) & 15)
#define MWB_META_STATIC(m) (((m) >> 8) & 1)
#define MWB_META_ALIVE(m) (((m) >> 9) & 1)
#define MWB_META_RADF32(m) (((m) >> 10) & 1)
#define MWB_MAX_ENTS 20   /* entity slots besides the agent (CollectHealth: 18) */

/* observation layouts */
enum { MWB_LAYOUT_HWC = 0,  /* [N,H,W,3]  MiniWorldEnv.observation_space, miniworld.py:473-478 */
       MWB_LAYOUT_CWH = 1 }; /* [N,3,W,H]  after TransposeImage, pytorch-a2c-ppo-acktr/envs.py:96-107 */

/* domain parameters in the order of the reference table params.py:110-123 */
enum { MWB_P_SKY_COLOR = 0, MWB_P_LIGHT_POS, MWB_P_LIGHT_COLOR, MWB_P_LIGHT_AMBIENT, MWB_P_OBJ_COLOR_BIAS,
       MWB_P_FORWARD_STEP, MWB_P_FORWARD_DRIFT, MWB_P_TURN_STEP, MWB_P_BOT_RADIUS, MWB_P_CAM_PITCH,
       MWB_P_CAM_FOV_Y, MWB_P_CAM_HEIGHT, MWB_P_CAM_FWD_DISP, MWB_NPARAM };

/* Constructor arguments: MiniWorldEnv.__init__ kwargs (miniworld.py:456-465) + the task class'
 * own kwargs + batch size / device. */
typedef struct mwb_config {
    int32_t abi_version;       /* MWB_ABI_VERSION */
    int32_t task;              /* MWB_TASK_* */
    int32_t num_envs;          /* environments owned by this handle (this GPU's shard) */
    int32_t obs_width;         /* 80  (miniworld.py:459) */
    int32_t obs_height;        /* 60  (miniworld.py:460) */
    int32_t want_depth;        /* also produce render_depth() maps (miniworld.py:1207-1220) */
    int32_t layout;            /* MWB_LAYOUT_* of the observation batch */
    int32_t domain_rand;       /* miniworld.py:464 */
    int32_t max_episode_steps; /* <= 0: the task's default (hallway.py:18, oneroom.py:14, ...) */
    int32_t device;            /* HIP device ordinal */
    double task_args[4];       /* see MWB_TASK_*; 0 = the class default */
    int32_t use_default_params; /* 1: params.py:110-123 DEFAULT_PARAMS; 0: `params` below */
    int32_t no_auto_reset;     /* 0: a finished env is reset inside mwb_step (VecEnv worker, subproc_vec_env.py:10-13);
                                  1: plain Gym semantics, the caller resets (miniworld.py:658-716) */
    double params[MWB_NPARAM][9]; /* default[3], min[3], max[3] per parameter (DomainParams.set) */
} mwb_config;

typedef struct mwb_handle mwb_handle;

/* library-owned device buffers produced by mwb_reset / mwb_step / mwb_render */
typedef struct mwb_outputs {
    uint8_t *obs;       /* u8, layout per config: [N,H,W,3] or [N,3,W,H]                         */
    float *depth;       /* f32 [N,H,W] metres (get_depth_map, opengl.py:336-371) or NULL         */
    float *reward;      /* f32 [N]  (VecPyTorch.step_wait casts to float32, envs.py:129)          */
    double *reward64;   /* f64 [N]  the reference's Python float, for bit-exact checks            */
    uint8_t *done;      /* u8  [N]                                                                */
    int32_t *ep_steps;  /* i32 [N]  step_count of the transition just taken (before auto-reset)   */
    size_t obs_bytes, depth_bytes;
    void *stack;        /* frame stack [N, nstack*3, W, H], f32 or u8 (mwb_stack_enable), or NULL              */
    size_t stack_bytes;
    float *feature;     /* f32 [N][2] info['feature'] of the transition (tmaze.py:311-318); zeros for other tasks */
    double *goal_pos;   /* f64 [N][3] info['goal_pos'] of the transition, T-maze family (tmaze.py:66,206); else zeros */
    /* reward64, goal_pos, reward, feature, ep_steps and done live in ONE allocation [pack, pack + pack_bytes) so that a
     * host-side consumer (the VecEnv contract returns numpy dones, CPU rewards and info dicts every step,
     * vec_env/subproc_vec_env.py:69-75) fetches them with a single device-to-host copy; each pointer above = pack + offset.  Order inside the allocation: done,
     * reward, feature, goal_pos, ep_steps, reward64 - a consumer that needs only the first few copies that prefix */
    void *pack;
    size_t pack_bytes;
} mwb_outputs;

/* host-side snapshot of the simulator state of a contiguous env range, for tests / debugging / checkpoints;
 * every pointer may be NULL to skip that field. B = mwb_num_boxes(h); boxes in the order of the reference's
 * entity list (hallway.py:31 one red box; tmaze.py:166-169 red, blue; simtorealpush.py:88-99 red, yellow). */
typedef struct mwb_state {
    double *agent_pos;  /* [count][3] Entity.pos (entity.py:22-24), y = 0 */
    double *agent_dir;  /* [count]    Entity.dir                          */
    double *box_pos;    /* [count][B][3]  y > 0 while the box is carried (miniworld.py:603-604) */
    double *box_dir;    /* [count][B]    */
    double *box_color;  /* [count][B][3] Box.color_vec (entity.py:381-383)   */
    double *box_size;   /* [count][B] Box.size[0] (entity.py:366-378): 0.8 unless the task draws it per episode */
    double *cam;        /* [count][4] cam_height, cam_fwd_disp, cam_pitch, cam_fov_y (entity.py:486-492) */
    double *sky_color, *light_pos, *light_color, *light_ambient; /* [count][3] (miniworld.py:561-566) */
    int32_t *step_count;    /* [count] miniworld.py:539,663 */
    int32_t *rng_pos;       /* [count] MT19937 position (RandomState.get_state()[2]); read-only */
    uint32_t *rng_keysum;   /* [count] sum of the 624 key words mod 2^32; read-only */
    int32_t *n_rooms;       /* [count] read-only */
    int32_t *n_segs;        /* [count] read-only */
    int32_t *goal_idx;      /* [count] current_goal (tmaze.py:91) / goal_box_idx (tmaze.py:135) */
    int64_t *episode_count; /* [count] tmaze.py:80,130 (1 after construction: MiniWorldEnv.__init__ resets once) */
    int64_t *task_step_count; /* [count] tmaze.py:240 */
    double *goal_dist;      /* [count] simtorealpush.py:84 */
    uint32_t *rng_state;    /* [count][625] RandomState.get_state(): the 624 key words, then the position */
    int32_t *carrying;      /* [count] agent.carrying as an index into the boxes, or -1 (miniworld.py:682-702) */
    /* general entity list (tasks >= MWB_TASK_PICKUPOBJS; B = mwb_num_boxes(h) slots = the episode's first list, agent excluded) */
    int32_t *ent_meta;      /* [count][B] see MWB_META_*; read-only */
    double *ent_radius;     /* [count][B] Entity.radius (a float32 value for meshes under NumPy >= 2); read-only */
    double *ent_height;     /* [count][B] Entity.height; read-only */
    double *ent_scale;      /* [count][B] MeshEnt.scale (0 for other kinds); read-only */
    int32_t *ent_order;     /* [count][B + 1] self.entities now, as slots: -2 = the agent, -1 = past the end (entities removed) */
    double *task_f;         /* [count] CollectHealth.health */
    int32_t *task_i;        /* [count] PickupObjs.num_picked_up */
    int32_t *text_tex;      /* [count][8] texture slot per character of the task's TextFrame (-1 = none / space); read-only */
} mwb_state;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* replaces: gym.make('MiniWorld-<Class>-v0') x N + SubprocVecEnv.__init__
 * (envs/__init__.py:43-49, pytorch-a2c-ppo-acktr/envs.py:57-63, vec_env/subproc_vec_env.py:36-56) */
int mwb_create(const mwb_config *cfg, mwb_handle **out);
/* replaces: SubprocVecEnv.close (vec_env/subproc_vec_env.py:87-97) */
int mwb_destroy(mwb_handle *h);
const char *mwb_last_error(void);
int mwb_abi_version(void);

/* ---- assets ------------------------------------------------------------------------------ */
/* replaces: Texture.load (opengl.py:71-108): upload RGB8 pixels (row 0 = TOP of the image, tightly
 * packed) for texture slot `tex_id` (0 floor_tiles_bw_1, 1-4 concrete_1..4, 5 concrete_tiles_1,
 * 6 brick_wall_1; the sim-to-real rinks also use 7-10 cardboard_1..4, 11-12 wood_1..2, 13 wood_planks_1,
 * 14 drywall_1, 15 stucco_1, 16 ceiling_tiles_1); the library builds the mip chain.  Every slot the task can
 * draw must be set before the first render. Synchronous. */
/* entity tasks: 17 asphalt_1, 18 slime_1, 19 cinder_blocks_1, 20 logo_mila_1, 21-24 the images of the textured meshes (medkit,
 * duckie, building, cone), 25 + 9 c + v: variant v + 1 of chars/ch_0x<ord>_*.png for the c-th character of "BLUERDGN" (Sign) */
#define MWB_NUM_TEXTURES 97
#define MWB_TEX_MESH0 21
#define MWB_TEX_CHAR0 25
int mwb_num_textures(mwb_handle *h);   /* how many leading slots the handle's task uses (7, 17, 25 or 97) */
/* replaces: ObjMesh.get / ObjMesh.__init__ (objmesh.py:16-216) for one mesh geometry, shared by all envs of the handle: the
 * triangle soup as the reference hands it to OpenGL - verts / norms float32 [n_tris][3][3], texcs [n_tris][3][2], in draw order -
 * its extents (ObjMesh.min_coords / max_coords), the texture slot of its image (-1 none) and a threaded bounding-volume
 * hierarchy over the triangles for the render kernel: nodes float32 [n_orders][n_nodes][8] = lo.xyz, skip (int bits) | hi.xyz,
 * first | count << 24 (int bits) in depth-first order (an inner node's first child follows it; `skip` = next node when the box
 * is missed or the leaf is done); perm int32 [n_tris] = triangle indices in leaf order.  n_orders is 1, or 8: eight threadings
 * of the same tree over the same leaves, threading k for rays whose direction (in the mesh's frame) has component a negative
 * iff bit a of k is set - the child nearer along the split axis first.  Host pointers; synchronous.
 * The vertex colour is the material's Kd, which the entity carries (ball_<c> / key_<c> share their geometry). */
int mwb_set_mesh(mwb_handle *h, int geom, int n_tris, const float *verts, const float *norms, const float *texcs, int tex_slot,
                 const float *min_coords, const float *max_coords, int n_nodes, int n_orders, const float *nodes, const int32_t *perm);
/* replaces: MeshEnt.__init__'s arithmetic (entity.py:118-127) - `scale = height / sy`, `radius = sqrt(sx^2 + sz^2) * scale` -
 * evaluated by the host with the reference's expressions, because their scalar type follows the installed NumPy (float32 under
 * NumPy >= 2, where sums with Python floats are then float32 too): one call per (geometry, height) the task builds. */
int mwb_set_mesh_dims(mwb_handle *h, int geom, double height, double scale, double radius, int is_float32);
/* debugging aid (MWB_DEBUG bit 4 at mwb_create): start / end s_memrealtime ticks (100 MHz) of every workgroup of the
 * last bulk render launch, [2 * n] u64; returns n or a negative code */
int mwb_debug_wg_times(mwb_handle *h, unsigned long long *out, int max_wgs);
/* debugging aid (MWB_EXP bit 2 at mwb_create): counters of the entity render kernel's mesh walks since the last reset - sample rays
 * entering the walk, walks started, node visits, triangle tests, wave loop iterations, wave calls (out8: 8 x u64) */
int mwb_debug_counters(mwb_handle *h, unsigned long long *out8, int reset);
int mwb_set_texture(mwb_handle *h, int tex_id, int width, int height, const uint8_t *rgb);

/* ---- simulation -------------------------------------------------------------------------- */
/* replaces: env.seed(seed + rank) per worker (envs.py:35-36, miniworld.py:528-530, random.py:9-10).
 * seeds: host array [num_envs]. Synchronous (host-side key hashing + upload). */
int mwb_seed(mwb_handle *h, const uint64_t *seeds);
/* the seed -> MT19937 init_by_array key of gym<=0.21's seeding.np_random (hash_seed: first 8 bytes of
 * sha512(str(seed)) as little-endian 32-bit limbs), host-only helper; returns the key length (1 or 2) */
int mwb_seed_key(uint64_t seed, uint32_t key[2]);
/* replaces: VecEnv.reset (vec_env/subproc_vec_env.py:77-80 -> MiniWorldEnv.reset miniworld.py:532-592)
 * mask: device u8[num_envs] or NULL; NULL or mask[i]!=0 resets env i. Renders the first observation. */
int mwb_reset(mwb_handle *h, const uint8_t *mask_dev, void *stream);
/* replaces: VecEnv.step_async+step_wait (vec_env/subproc_vec_env.py:58-75, worker 5-14,26-31 ->
 * MiniWorldEnv.step miniworld.py:658-716 + task rule e.g. envs/maze.py:106-113), including the
 * worker's auto-reset and the fork's `mask` ('dummy') semantics.
 * actions: device i32[num_envs], MiniWorldEnv.Actions values 0..7 (miniworld.py:437-454: turn_left, turn_right, move_forward,
 * move_back, pickup, drop, toggle, done - every task executes all of them as the base class does; any other value does
 * nothing but count a step, as the reference's if-chain does); skip_mask: device u8[num_envs] or NULL (mask[i]!=0 -> env i is not
 * stepped, reward -99, done 0, observation re-rendered). */
int mwb_step(mwb_handle *h, const int32_t *actions_dev, const uint8_t *skip_mask_dev, void *stream);
/* the same with the LongTensor the reference's policy produces (VecPyTorch.step_async, envs.py:121-125): int64 actions [N] in
 * device memory are read in place (no conversion pass); a value outside int32 is an unknown action (never aliased) */
int mwb_step_i64(mwb_handle *h, const int64_t *actions_dev, const uint8_t *skip_mask_dev, void *stream);
/* replaces: MiniWorldEnv.render_obs / render_depth (miniworld.py:1160-1220) for the whole batch */
int mwb_render(mwb_handle *h, void *stream);
int mwb_get_outputs(mwb_handle *h, mwb_outputs *out);

/* ---- learner-side layout fusion ---------------------------------------- */
/* replaces: VecPyTorchFrameStack + VecPyTorch's .float() (pytorch-a2c-ppo-acktr/envs.py:117-165): a
 * library-owned channel-first stack [N, nstack*3, W, H] (needs MWB_LAYOUT_CWH), dtype 0 = u8, 1 = f32
 * (values 0..255).  mwb_stack_update(after_reset != 0) implements reset(): zero, newest obs in the last 3
 * channels (envs.py:158-162); after_reset == 0 implements step_wait(): shift by 3 channels, zero the envs
 * whose `done` is set, append the newest obs (envs.py:149-156) - one pass over the stack. */
int mwb_stack_enable(mwb_handle *h, int nstack, int dtype);
/* dtype | MWB_STACK_SLIDING: the stack as a sliding window over nstack*3 + 3*MWB_STACK_SLACK_FRAMES planes per env - a step
 * writes the new frame only (the history planes stay in place) and the window moves three planes on; once every
 * MWB_STACK_SLACK_FRAMES + 1 steps the history is copied back to the front.  The current view is planes
 * [first_plane, first_plane + nstack*3) of each env's planes_per_env (mwb_stack_window, valid after each mwb_stack_update):
 * same values as the shifting stack at ~1/4 of its HBM traffic.  The window position lives on the host, so a captured graph
 * of mwb_stack_update must not be replayed with this flag. */
#define MWB_STACK_SLIDING 16
#define MWB_STACK_SLACK_FRAMES 8
/* dtype | MWB_STACK_FUSED (implies the sliding window): mwb_reset / mwb_step move the window themselves and the render
 * kernels write each new frame straight into its newest three planes (u8 -> f32 on the way out of LDS), zeroing the history of
 * the envs they regenerate: mwb_stack_update becomes a no-op and the observation never takes a second trip through HBM.
 * Read the window with mwb_stack_window after every mwb_reset / mwb_step.  Must be enabled before the first mwb_reset /
 * mwb_step / mwb_render (MWB_ESTATE otherwise: the window would lack the frame already rendered).  A PARTIAL mwb_reset(mask)
 * is a step for the window: it moves three planes on, the masked envs get a zeroed history and their first frame, the others
 * get their re-rendered current frame appended once more (VecPyTorchFrameStack has no partial reset to compare with).
 * A view of an earlier window position is guaranteed only until the next pass: the history planes of an env that ends are
 * zeroed in place; for envs that keep running it stays intact for MWB_STACK_SLACK_FRAMES + 1 - nstack further steps. */
#define MWB_STACK_FUSED 32
int mwb_stack_window(mwb_handle *h, int *first_plane, int *planes_per_env);
int mwb_stack_update(mwb_handle *h, int after_reset, void *stream);

/* World generation cannot fail for the four tasks with sane arguments; if it ever does (a portal outside
 * its wall, more than one portal on an edge, a placement that finds no free spot in 100000 draws - the
 * reference would assert or spin) the kernel flags it. Synchronous: returns MWB_ESTATE if any env of the
 * handle hit such a condition since creation. */
int mwb_check(mwb_handle *h);

/* ---- introspection (tests, Gym single-env view) ------------------------------------------- */
int mwb_num_boxes(mwb_handle *h);   /* B: boxes per env of the handle's task */
int mwb_get_state(mwb_handle *h, int first_env, int count, mwb_state *out);          /* synchronous */
/* replaces: assigning env.agent.pos / env.box.pos / env.rand.np_random.set_state(...) etc. on the reference's
 * Python objects - overwrite the state of an env range with every non-NULL, non-read-only field of `in`
 * (test hook: inject oracle / reference states; also restores a snapshot taken with mwb_get_state). Synchronous.
 * MWB_EINVAL for non-finite values, box_size <= 0, carrying / goal_idx / MT19937 position out of range. */
int mwb_set_state(mwb_handle *h, int first_env, int count, const mwb_state *in);
/* overwrite pose / step counter of env range (NULL = leave); used to inject oracle states */
int mwb_set_agent(mwb_handle *h, int first_env, int count, const double *pos_xz, const double *dir,
                  const int32_t *step_count);
/* replaces: assigning `env.domain_rand = flag` after construction, as the reference's own smoke test does
 * for every env (run_tests.py:64-66; read at reset miniworld.py:558 and at every step miniworld.py:665);
 * takes effect from the next mwb_reset / mwb_step. Synchronous. */
int mwb_set_domain_rand(mwb_handle *h, int domain_rand);
/* overwrite the goal-alternation state of the T-maze family for an env range (NULL = leave); test hook */
int mwb_set_task_state(mwb_handle *h, int first_env, int count, const int64_t *episode_count,
                       const int64_t *task_step_count, const int32_t *goal_idx);
/* replaces: MiniWorldEnv.intersect(ent, pos, radius) (miniworld.py:933-959) for one env. `ent` = position of the
 * querying entity in the entity list (0 .. B-1 the boxes, B the agent) - it is skipped, as `ent2 is ent` is - or -1
 * to skip nobody.  result: 0 none, 1 wall (walls are tested first), 2 + k = the first entity k in list order within
 * radius + its radius. Synchronous. */
int mwb_intersect(mwb_handle *h, int env, int ent, double x, double z, double radius, int *result);
/* geometry of one env as the kernels see it: n_rooms x mwb_room_words(h) f32 words and n_segs x 4 f64.
 * Rectangle tasks: MWB_ROOM_WORDS words per room (min_x max_x min_z max_z, height, textures, neighbours, 4 x side).
 * MWB_TASK_YMAZE: MWB_POLY_ROOM_WORDS words per room - height, textures, n_edges | culled << 8, pad, then 4 edges of
 * 12 words: p.x p.z dir.x dir.z | n.x n.z portal_lo portal_hi | portal_max_y neighbour(int bits, -1 none) 0 0
 * (Room.outline / edge_dirs / edge_norms / portals, miniworld.py:75-218, as float32). */
#define MWB_ROOM_WORDS 24
#define MWB_POLY_ROOM_WORDS 52
int mwb_room_words(mwb_handle *h);
int mwb_get_geometry(mwb_handle *h, int env, float *rooms, int max_rooms, double *segs, int max_segs,
                     int *n_rooms, int *n_segs);

/* replaces: MiniWorldEnv.render_top_view(frame_buffer) (miniworld.py:1087-1158; `render(view='top')` 1317-1335) for the whole
 * batch: the floorplan from straight above, extents + 1 m widened to the frame's aspect, floors, box tops and the agent's
 * triangle.  out_dev: uint8 [N][height][width][3] in device memory, any frame size; enqueued on `stream`.  A display /
 * debugging view, not part of the step pipeline. */
int mwb_render_top_view(mwb_handle *h, uint8_t *out_dev, int width, int height, void *stream);

/* replaces: MiniWorldEnv.render_obs(frame_buffer) / render_depth(frame_buffer) with ANOTHER frame buffer than the observation's
 * (miniworld.py:1160-1220) - in particular render(mode='rgb_array', view='agent'), which draws the agent's view into the
 * 800 x 600 `vis_fb` (miniworld.py:505,1317-1335): the current state of every env at width x height, same render spec as the
 * observation (8 samples per pixel; the reference asks for 16 in vis_fb - not modelled).  out_dev: uint8
 * [N][height][width][3], depth_dev: float32 [N][height][width] metres or NULL; device memory, enqueued on `stream`.  The view is
 * rendered in tiles (one workgroup each), so its size is not bound by LDS.  Not part of the step pipeline. */
int mwb_render_view(mwb_handle *h, uint8_t *out_dev, float *depth_dev, int width, int height, void *stream);

/* replaces: MiniWorldEnv.get_visible_ents() (miniworld.py:1222-1315) for the whole batch: per env a bit mask over the boxes in
 * entity-list order, bit b set iff box b's 0.2 m query cube passes its occlusion query (any of the 8 x W x H samples of the
 * observation frame nearer than the rooms and than the cubes of the boxes before it).  mask_dev: uint32 [N] in device memory;
 * enqueued on `stream`.  The reference never calls the function and holds no output of it (parity unpinned). */
int mwb_visible_ents(mwb_handle *h, uint32_t *mask_dev, void *stream);

/* ---- timing hooks used by bench.py -------------------------------------------------------- */
/* average device time (ms) of each kernel of the step pipeline since the last call (HIP events
 * recorded on the stream the kernels were launched on); names: "step","reset","prep","render".
 * enable: 0 off, 1 every pass, n > 1 every n-th pass (the seven events of a pass cost ~20 us of stream time). */
int mwb_timing_enable(mwb_handle *h, int enable);
int mwb_timing_read(mwb_handle *h, double *ms_step, double *ms_reset, double *ms_prep, double *ms_render,
                    int *n_samples);

#ifdef __cplusplus
}
#endif
#endif
