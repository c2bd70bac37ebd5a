/* mw_oracle.h - CPU restatement ("oracle") of the gym-miniworld step + render_obs hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the
 * product (gym_miniworld_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status:
 *   - state half (RNG stream, world generation, placement, step, collision, reward/done,
 *     camera vectors): PINNED bit-exactly against vectors produced by the unmodified
 *     reference (tests/golden/state_*.npz, math_kat.npz; generator committed beside them).
 *   - seed -> MT19937 key (un-vendored `gym` <= 0.21, gym/utils/seeding.py, recalled):
 *     PARITY UNPINNED (no gym in the build container); isolated on the Python side.
 *   - pixel half (render_obs / render_depth): PARITY UNPINNED against the real OpenGL
 *     driver - no GL exists in this pipeline.  The renderer below is a frozen written
 *     specification of the fixed-function pipeline as the reference configures it
 *     (DESIGN.md "render spec"); its *inputs* (polygons, texcoords, normals, light, camera)
 *     are pinned against the reference's captured GL call stream (tests/golden/glstream_*.json).
 */
#ifndef MW_ORACLE_H
#define MW_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MWO_HALLWAY = 0, MWO_ONEROOM = 1, MWO_FOURROOMS = 2, MWO_MAZE = 3, MWO_TMAZE = 4, MWO_TMAZE_TWOBOX = 5,
       MWO_SIM2REAL_GOTO = 6, MWO_SIM2REAL_PUSH = 7, MWO_PUTNEXT = 8, MWO_YMAZE = 9,
       /* tasks with a general entity list (mesh entities, frames; entities that leave or re-enter the list) */
       MWO_PICKUPOBJS = 10, MWO_ROOMOBJS = 11, MWO_COLLECTHEALTH = 12, MWO_THREEROOMS = 13, MWO_SIGN = 14,
       MWO_SIDEWALK = 15, MWO_WALLGAP = 16, MWO_NTASKS };

/* entity kinds (reference entity.py): Box 362-408, MeshEnt / Key / Ball 100-146, 410-434, ImageFrame 148-242, TextFrame 244-360 */
enum { MWO_ENT_BOX = 0, MWO_ENT_MESH = 1, MWO_ENT_IMAGE = 2, MWO_ENT_TEXT = 3 };
/* mesh geometries: gym_miniworld/meshes/<name>.obj; ball_<c> / key_<c> share one geometry each, the colour is the entity's */
enum { MWO_MESH_BALL = 0, MWO_MESH_KEY, MWO_MESH_MEDKIT, MWO_MESH_DUCKIE, MWO_MESH_BUILDING, MWO_MESH_CONE, MWO_NMESH };

/* domain parameters, reference params.py:110-123, same order */
enum {
    MWO_P_SKY_COLOR = 0, MWO_P_LIGHT_POS, MWO_P_LIGHT_COLOR, MWO_P_LIGHT_AMBIENT, MWO_P_OBJ_COLOR_BIAS,
    MWO_P_FORWARD_STEP, MWO_P_FORWARD_DRIFT, MWO_P_TURN_STEP, MWO_P_BOT_RADIUS, MWO_P_CAM_PITCH,
    MWO_P_CAM_FOV_Y, MWO_P_CAM_HEIGHT, MWO_P_CAM_FWD_DISP, MWO_NPARAM
};

#define MWO_MAX_BOXES 20   /* entity slots besides the agent (CollectHealth has 18) */
#define MWO_MAX_ROOMS 512
#define MWO_MAX_PORTALS 2 /* per edge */
#define MWO_MAX_TEX 128
#define MWO_MAX_LEVELS 12

typedef struct MwoEnv MwoEnv;

/* textures: global table shared by all envs; data = RGBA8 rows bottom-up (GL convention),
 * level l at data + level_off[l], dims max(1, w>>l) (floor halving) */
int mwo_set_texture(int tex_id, int width, int height, int n_levels, const uint8_t *rgba_all_levels);

/* task_args: Hallway {length}, OneRoom {size}, FourRooms {}, Maze {num_rows, num_cols, room_size},
 * TMaze {goal_pos given?, goal x, goal z, sub_task_length (> 0: TMazeDynamic)},
 * TMazeTwoBox {rule: 0 episode count (TMazeTwoBoxDynamic) / 1 step count + features (*Features*), -, -, sub_task_length},
 * SimToRealGoTo {} / SimToRealPush {} (envs/simtorealgoto.py, simtorealpush.py: pass their sim_params table and domain_rand = 1),
 * PutNext {size} (envs/putnext.py: six boxes in COLOR_NAMES order - blue green grey purple red yellow - of random sizes; actions 0..7
 * incl. pickup 4 / drop 5 with the carry physics of miniworld.py:594-702; done when red is next to yellow and nothing is carried).
 * PickupObjs {size, num_objs}, RoomObjs {size}, CollectHealth {size}, ThreeRooms {}, Sign {size, color_index, goal},
 * Sidewalk {}, WallGap {} (envs/pickupobjs.py, roomobjs.py, collecthealth.py, threerooms.py, sign.py, sidewalk.py, wallgap.py).
 * params: MWO_NPARAM x 9 doubles (default[3], min[3], max[3]); NULL = reference defaults. */
MwoEnv *mwo_create(int task, const double *task_args, int max_episode_steps, int domain_rand,
                   const double *params);
void mwo_destroy(MwoEnv *e);

void mwo_seed_key(MwoEnv *e, const uint32_t *key, int key_len); /* RandomState.seed(list) */
void mwo_reset(MwoEnv *e);
void mwo_step(MwoEnv *e, int action, double *reward, int *done); /* no auto-reset */

/* state access */
typedef struct {
    double agent_pos[3], agent_dir;
    double box_pos[3], box_dir, box_color[3];
    double cam_height, cam_fwd_disp, cam_pitch, cam_fov_y;
    double sky_color[3], light_pos[3], light_color[3], light_ambient[3];
    double cam_pos[3], cam_dir[3];
    int step_count, max_episode_steps, n_rooms, n_segs, n_quads;
    int rng_pos;
    uint32_t rng_key0, rng_key1, rng_key623, rng_keysum;
    /* T-maze family */
    int n_boxes, goal_idx;
    double box2_pos[3], box2_dir, box2_color[3];
    long long episode_count, task_step_count;
    double feature[2]; /* info['feature'] of the last step (zeros for tasks without) */
    /* sim-to-real tasks: per-episode sizes */
    double box_size, box2_size, agent_radius, goal_dist;
    /* all boxes in entity-list order (y = pos[1] > 0 while carried), and agent.carrying as a box index or -1 */
    double boxes_pos[MWO_MAX_BOXES][3], boxes_dir[MWO_MAX_BOXES], boxes_color[MWO_MAX_BOXES][3], boxes_size[MWO_MAX_BOXES];
    int carrying;
    /* general entity list: per slot (= position in the episode's first list) kind, mesh geometry, flags, radius / height / scale;
     * `order` = the entity list now, as slots (-2 = the agent), n_order entries */
    int ents_kind[MWO_MAX_BOXES], ents_mesh[MWO_MAX_BOXES], ents_alive[MWO_MAX_BOXES], ents_static[MWO_MAX_BOXES], ents_rad_f32[MWO_MAX_BOXES];
    double ents_radius[MWO_MAX_BOXES], ents_height[MWO_MAX_BOXES], ents_scale[MWO_MAX_BOXES];
    int order[MWO_MAX_BOXES + 1], n_order;
    double health; int num_picked;
    int ents_tex[MWO_MAX_BOXES][8];   /* ImageFrame: [0] = texture id; TextFrame: one id per character (-1 = space) */
} MwoState;

/* meshes: global table like the textures.  Arrays as objmesh.py builds them (float32, draw order): verts / norms [n][3][3],
 * texcs [n][3][2]; tex_id = texture slot of the mesh's image or -1.  The vertex colour (Kd) is the entity's colour. */
int mwo_set_mesh(int geom, int n_tris, const float *verts, const float *norms, const float *texcs, int tex_id,
                 const float *min_coords, const float *max_coords);
/* MeshEnt.__init__ (entity.py:118-127) evaluated by the caller with the reference's expressions and the installed NumPy's
 * scalar types: scale and radius of geometry `geom` at `height`, and whether they are float32 scalars (NumPy >= 2) */
int mwo_set_mesh_dims(int geom, double height, double scale, double radius, int is_f32);
void mwo_get_state(MwoEnv *e, MwoState *out);
void mwo_set_agent(MwoEnv *e, double x, double z, double dir); /* test hook */
void mwo_set_step_count(MwoEnv *e, int step_count);
void mwo_set_box(MwoEnv *e, int box /* index in the entity list */, double x, double z, double dir); /* test hook */
void mwo_set_box_y(MwoEnv *e, int box, double y);
void mwo_set_carrying(MwoEnv *e, int box /* or -1 */);
void mwo_render_step_frame(MwoEnv *e, int on); /* 1: render the entities as the last step's own frame saw them (before the task rule) */
int mwo_intersect_ent(MwoEnv *e, int ent_index /* -1 nobody, 0..B-1 box, B agent */, double x, double z, double radius);
void mwo_set_counters(MwoEnv *e, long long episode_count, long long task_step_count, int goal_idx); /* test hook */
/* geometry dumps (sizes from MwoState): outline R*4*2, heights R, portals R*4*MAXP*4 (nan pad),
 * portal_count R*4, segs S*4 (a.x a.z b.x b.z), room_probs R, quad verts Q*4*3, norms Q*4*3,
 * texcs Q*4*2 (float), quad_offsets R+1, floor_texcs R*4*2, ceil_texcs R*4*2, tex ids R*3 */
void mwo_get_geometry(MwoEnv *e, double *outline, double *heights, double *portals, int *portal_count,
                      double *segs, double *room_probs, double *quad_verts, double *quad_norms,
                      float *quad_texcs, int *quad_offsets, double *floor_texcs, double *ceil_texcs,
                      int *tex_ids);
int mwo_intersect(MwoEnv *e, int ent /*0=box,1=agent,2=box2*/, double x, double z, double radius);

/* render the agent's view (reference render_obs / render_depth); rgb: H*W*3 row 0 = top,
 * depth: H*W float metres or NULL */
void mwo_render(MwoEnv *e, int width, int height, uint8_t *rgb, float *depth);
/* render_top_view(frame_buffer) (miniworld.py:1087-1158): the map from above, agent drawn, at any frame size */
void mwo_render_top(MwoEnv *e, int width, int height, uint8_t *rgb);
/* get_visible_ents() (miniworld.py:1222-1315): bit b = box b passes its occlusion query at the observation size */
uint32_t mwo_visible_ents(MwoEnv *e, int width, int height);

/* stand-alone pieces with reference known-answer vectors */
int mwo_intersect_circle_segs(const double *pt3, double radius, const double *segs /*n*2*3*/, int n);
void mwo_gen_rot_matrix(const double *axis3, double angle, double *out9);

/* timed loop for bench.py's cpu_baseline: n_steps of step(+auto reset)+render with the
 * counter-based action stream; returns wall seconds */
double mwo_bench_loop(MwoEnv *e, int n_steps, uint64_t action_seed, uint64_t env_index, int width,
                      int height, int want_depth, int constant_action /* -1 = random */, int n_actions /* random actions over range(n), 0 = 3 */);

#ifdef __cplusplus
}
#endif
#endif
