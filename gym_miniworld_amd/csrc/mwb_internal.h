// mwb_internal.h - data layout shared by the host API (mwb_api.hip) and the kernels (mwb_kernels.hip).
//
// HBM layout (one handle = one GPU's shard of N environments), structure-of-arrays:
//   sim  : per-env f64 pose / episode parameters (arrays of N), i32 counters, u8 flags
//   rng  : per-env MT19937 state, N x 625 u32 (624 key words + position), env-major
//   rooms: per-env room table, N x R_max x MWB_ROOM_WORDS (24) f32 words (render kernel stages it in LDS)
//   segs : collision segments, S_max x 4 x N f64 (a.x a.z b.x b.z; segment-major, env-minor so that
//          one-env-per-lane reads coalesce), reference order per env
//   frame: per-env render constants (camera basis, lit colours, box frames), N x frame_words f32
//   tex  : RGBA8 mip pyramids of the 7 textures, shared by all envs (L2 / Infinity Cache resident)
//   out  : obs u8 [N,H,W,3] or [N,3,W,H], depth f32 [N,H,W], reward f32/f64 [N], done u8 [N]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/miniworld_batch.h"

#define MWB_MT_WORDS 625   // 624 key words + pos
#define MWB_MAX_TEX MWB_NUM_TEXTURES
#define MWB_MAX_LEVELS 12
#define MWB_MAX_BOXES 6   // box-only tasks (templated step / render kernels); entity tasks: MWB_MAX_ENTS slots
#define MWB_RESET_MAX_BLOCKS 4096   // largest grid reset_kernel is launched with
#define MWB_ENT_AGENT 255 // the agent's entry in an env's entity order list
#define MWB_ORDER_STRIDE 24   // bytes per env of d.ent_order (MWB_MAX_ENTS + 1 entries, padded)
// frame constants: 36 fixed words, then one block of FC_BOX_STRIDE words per box; d.frame_words = that, rounded up to 4

// room table: 24 f32 words (96 B) per room; ints stored as bit patterns
//   0-3  min_x max_x min_z max_z
//   4    wall height          5  textures wall | floor << 8 | ceil << 16, bits 24-27: wall-u runs backwards on side s
//   6    neighbour room behind the portal of side 0 | side 1 << 16 (0xFFFF = no portal)     7  sides 2 | 3 << 16
//   8+4s lo hi max_y u_org of side s (0:+x east, 1:-z north, 2:-x west, 3:+z south): portal extent along the
//        side's axis and in y (min_y is 0 in every supported task - enforced by reset_kernel), origin of
//        the wall texture's u coordinate
#define RW_MINX 0
#define RW_MAXX 1
#define RW_MINZ 2
#define RW_MAXZ 3
#define RW_HEIGHT 4
#define RW_TEX 5
#define RW_NBR01 6
#define RW_NBR23 7
#define RW_SIDE0 8
#define RW_SIDE_WORDS 4
#define RS_LO 0
#define RS_HI 1
#define RS_MAXY 2
#define RS_UORG 3
#define RW_NO_NBR 0xFFFFu

// polygon room table (MWB_TASK_YMAZE): MWB_POLY_ROOM_WORDS = 52 f32 words per room
//   0 wall height   1 textures (as RW_TEX, no flip bits)   2 n_edges | culled << 8   3 pad
//   4 + 12 k, edge k (a missing 4th edge has n = (0, 0): never an exit, never excludes the eye):
//     p.x p.z dir.x dir.z | n.x n.z lo hi | max_y nbr 0 0   nbr: room behind the portal as int bits, -1 none; a portal into a
//     "culled" connector (reversed winding, nothing of it is drawn) points on to the room behind the connector
#define PW_HEIGHT 0
#define PW_TEX 1
#define PW_FLAGS 2
#define PW_EDGE0 4
#define PW_EDGE_WORDS 12

// frame-constant word offsets
#define FC_EYE 0
#define FC_F 3
#define FC_S 6
#define FC_U 9
#define FC_TW 12
#define FC_TH 13
#define FC_SKY 14
#define FC_LIT_FLOOR 17
#define FC_LIT_CEIL 20
#define FC_LIT_WALL 23   // 4 x 3
#define FC_LIGHT_DIR 23  // polygon rooms (d.poly): the light itself instead of the four axis-aligned wall colours -
#define FC_LIGHT_AMB 26  //   a wall's colour is computed from its edge normal when it is shaded
#define FC_LIGHT_DIF 29
#define FC_BOX_IN_VIEW 35      // != 0 if some box's footprint-inflated sphere can meet the view cone
// per-box block of FC_BOX_STRIDE words starting at FC_LIT_BOX; box b at + b * FC_BOX_STRIDE
#define FC_BOX_STRIDE 34
#define FC_LIT_BOX 36    // 6 x 3
#define FC_BOX_POS 54
#define FC_BOX_C 57
#define FC_BOX_S 58
#define FC_BOX_HX 59
#define FC_BOX_HZ 60
#define FC_BOX_SY 61
#define FC_BOX_LO 62     // ray origin in box-local axes (3)
#define FC_CULL_OC 65    // box bounding-sphere centre minus eye (3)
#define FC_CULL_CC 68    // |oc|^2 - R^2
#define FC_CULL_CC_PIXEL 69   // same with R grown by a pixel footprint
#define MWB_FRAME_WORDS_FOR(n_boxes) ((36 + FC_BOX_STRIDE * (n_boxes) + 3) & ~3)
// Entity tasks: the same FC_BOX_STRIDE-word block per entity slot, read by kind.  FC_BOX_HX >= 0: a box (as above); -1: a mesh;
// -2: an image / text frame; an entity that has left the list has FC_CULL_CC = FC_CULL_CC_PIXEL = +inf (no ray ever passes its gate).
//   mesh  (offsets from the block's start = FC_LIT_BOX - 36): 0-2 light direction in the mesh's frame / scale, 3-5 Kd,
//         6-8 0.2 Kd + ambient Kd, 9-11 diffuse, 12 1 / scale, 13 geometry (int), 14 texture slot (int, -1 none), 15-17 + FC_BOX_SY: gate box;
//         FC_BOX_POS, FC_BOX_C, FC_BOX_S as for a box; FC_BOX_LO = the eye in the mesh's frame (rotated, / scale)
//   frame: 0-2 lit colour of the front, 3 depth, 4 half height, 5 half width, 6 character width, 7 characters (int),
//         8-15 texture slot per character (int, -1 = blank); FC_BOX_POS, C, S; FC_BOX_LO = the eye in the frame's axes
#define FE_MESH_LL 0
#define FE_MESH_KD 3
#define FE_MESH_AMB 6
#define FE_MESH_DIF 9
#define FE_MESH_INVS 12
#define FE_MESH_GEOM 13
#define FE_MESH_TEX 14
#define FE_MESH_BPAD 15   // gate box (below) grown by a pixel's footprint at the mesh's far side, in mesh units
#define FE_MESH_BHX 16    // a conservative box of the mesh in its own frame, for the interior-pixel gate: |x| <= BHX, |z| <= BHZ,
#define FE_MESH_BHZ 17    //   0 <= y <= FC_BOX_SY (objmesh.py re-centres every mesh: base at y = 0, x and z about the middle)
#define FE_FRAME_LIT 0
#define FE_FRAME_SX 3
#define FE_FRAME_HY 4
#define FE_FRAME_HZ 5
#define FE_FRAME_CW 6
#define FE_FRAME_NCH 7
#define FE_FRAME_TEX 8

// one mesh geometry in HBM (shared by all envs; L2 resident): BVH nodes (2 float4 each), triangle records in LEAF order
// (3 float4: v0.xyz e1.x | e1.yz e2.xy | e2.z, original index (int bits), 0, 0) and shading records by ORIGINAL index
// (4 float4: n0.xyz n1.x | n1.yz n2.xy | n2.z tc0.st tc1.s | tc1.t tc2.st 0)
struct MwbMeshDesc {
    int n_tris, n_nodes, tex_id;
    int n_orders;   // threadings of the hierarchy stored one after the other at node_off: 1, or 8 = one per sign pattern of the ray direction
    uint32_t node_off, tri_off, shade_off, tri2_off;   // float4 offsets into d.mesh_data; tri2 = the triangle records by ORIGINAL index (shading)
    float min_c[4], max_c[4];
};
struct MwbMeshDims { int geom, is_f32; double height, scale, radius; };
#define MWB_MAX_MESH_DIMS 16

struct MwbTexDesc {
    int w, h, n_levels;
    float sc_s, sc_t;                     // TEX_DENSITY / size (miniworld.py:17,30-31,58-63) as f32
    int pad[3];
    uint32_t level_off[MWB_MAX_LEVELS];   // texel (u32) offsets into the shared texel buffer
};

struct MwbParam { double def[3], lo[3], hi[3]; };

// everything a kernel needs, passed by value
struct MwbDev {
    int N, task, W, H, want_depth, layout, domain_rand, max_episode_steps;
    int R_max, S_max, auto_reset;
    int n_boxes;       // 1; 2 for MWB_TASK_TMAZE_TWOBOX (box 0 red, box 1 blue) and SIM2REAL_PUSH (red, yellow); 6 for PUTNEXT (COLOR_NAMES order)
    int frame_words;   // MWB_FRAME_WORDS_FOR(n_boxes)
    int n_tex;         // leading texture slots the task can draw (7, or MWB_NUM_TEXTURES for the sim-to-real rinks)
    int split_envs;    // the last split_envs envs of a bulk render launch are rendered by two half-frame workgroups each
    int act_stride;    // int32 words between consecutive envs' actions: 1, or 2 when the caller hands over int64 actions
    int poly;          // rooms are general convex polygons (YMaze): polygon room table, POLY render kernels
    int room_words;    // f32 words per room of d.rooms: MWB_ROOM_WORDS or MWB_POLY_ROOM_WORDS
    int no_ceiling;    // the task's rooms have no ceiling (sim-to-real rinks): selects the NOCEIL render kernels
    double agent_radius;   // entity.py:451 (0.4), 0.11 in the sim-to-real rinks
    int debug_flags;   // MWB_DEBUG env var at mwb_create: bit0 = resolve every pixel with the full 8-sample path
    double task_args[4];
    MwbParam params[MWB_NPARAM];
    // sim state (f64 SoA)
    double *agent_x, *agent_z, *agent_dir;
    double *box_x, *box_z, *box_dir;   // [n_boxes][N]
    double *box_y;          // [n_boxes][N] 0 on the floor, the carry height while carried (miniworld.py:603-604)
    int32_t *carrying;      // [N] agent.carrying as a box index, or -1 (miniworld.py:682-702)
    double *box_color;      // [n_boxes][N][3]
    double *box_size;       // [n_boxes][N] Box edge length (0.8 unless the task draws it per episode)
    double *goal_dist;      // [N] SimToRealPush
    int64_t *episode_count, *task_step_count;   // [N] goal-alternation counters of the T-maze family (envs/tmaze.py)
    int32_t *goal_idx;      // [N]
    double *cam;            // [N][4] height, fwd_disp, pitch, fov_y
    double *sky_color, *light_pos, *light_color, *light_ambient;   // [N][3]
    int32_t *step_count, *n_rooms, *n_segs;
    double *seg_stage;   // reset_kernel's staging of an env's collision segments, MWB_RESET_MAX_BLOCKS x S_max x 4 (one row set per block)
    int32_t *n_rrooms;   // records in the env's room table: n_rooms, + 1 where a room with two openings on one wall was cut in two (reset_kernel)
    int32_t *error_flag;    // [1] set by reset_kernel when world generation hits a condition the reference asserts on
    uint8_t *reset_set;     // which envs are (re)generated in the current pass (set by step / mark_reset)
    int32_t *reset_list;    // [N] the same envs as a compact list (order arbitrary) ...
    int32_t *reset_count;   // [1] ... and its length; zeroed by clear_list_kernel at the end of every pass
    uint32_t *rng;          // [N][625]
    float *rooms;           // [N][R_max][room_words]
    double *segs;           // [S_max][4][N]
    float *frame;           // [N][frame_words]
    void *stk;              // fused frame stack (MWB_STACK_FUSED) or null: base of [N][stk_K][W][H] planes, f32 (stk_float) or u8;
    int stk_float, stk_C, stk_K, stk_pos;   // the window the frame being rendered belongs to starts at plane stk_pos
    double *world_ext;      // [N][4] min_x max_x min_z max_z of the floorplan (miniworld.py:576-579), written by reset_kernel
    const uint32_t *texels;
    const MwbTexDesc *tex_desc;   // [MWB_MAX_TEX] in device memory
    // outputs
    uint8_t *obs;
    float *depth, *reward;
    double *reward64;
    uint8_t *done;
    int32_t *ep_steps;
    uint32_t *cost;         // [2N] s_memrealtime ticks (10 ns) the last bulk render spent on env e: whole frame at [2e], or the halves
    uint8_t *bucket;        // [N] scratch of order_kernel
    // blockIdx -> env maps of the bulk render, envs by decreasing measured frame cost (slow frames first), double-buffered
    // ON THE DEVICE so that a captured hipGraph keeps the whole mechanism: order_state[0] selects the map in use,
    // order_kernel fills the other one and raises order_state[1]; the next step_kernel flips the selection.
    int32_t *order_bufs[2]; // [N] each
    int32_t *order_state;   // [2] {map in use, the other map is complete}
    unsigned long long *wg_ts;   // [2 * (N + split_envs)] start / end s_memrealtime of every bulk render workgroup, or null (MWB_DEBUG bit 4)
    float *feature;         // [N][2]
    double *goal_pos;       // [N][3]
    // ---- entity tasks (task >= MWB_TASK_PICKUPOBJS): the general entity list.  Slot arrays are [n_boxes][N] like the box arrays
    // (box_x/y/z/dir = position and heading of any entity, box_size = Box edge / MeshEnt height, box_color = Box colour / mesh Kd)
    int ent_task;           // 1 for those tasks: step_ents_kernel, kind-aware prep, the ENT render kernels
    int step_pass;          // the frame being prepared is a step's own (an entity the task rule removed / respawned is still drawn)
    int32_t *ent_meta;      // [n_boxes][N] MWB_META_* word
    double *ent_radius, *ent_height, *ent_scale;   // [n_boxes][N]
    uint8_t *ent_order;     // [N][MWB_ORDER_STRIDE] self.entities as slots (MWB_ENT_AGENT = the agent), n_order entries
    int32_t *n_order;       // [N]
    double *task_f;         // [N] CollectHealth.health
    int32_t *task_i;        // [N] PickupObjs.num_picked_up
    int32_t *text_tex;      // [N][8] texture slot per character of the TextFrame
    int32_t *ovr_slot;      // [N] slot whose pose the step's frame takes from ovr_pose (-1 none): the entity as the frame saw it
    double *ovr_pose;       // [N][4] x y z dir
    const MwbMeshDesc *mesh_desc;   // [MWB_NUM_MESHES] in device memory
    const float4 *mesh_data;
    MwbMeshDims mesh_dims[MWB_MAX_MESH_DIMS];
    int n_mesh_dims;
    int tile_w, tile_h;   // entity tasks: every frame is rendered as tiles of this size, one workgroup each (0: whole frames)
    unsigned long long *dbg_counters;   // [8] or null: walk_meshes counters (MWB_EXP bit 2)
    int exp_flags;   // MWB_EXP env var at mwb_create: bit 0 = meshes are never hit, bit 1 = flat grey mesh shading (timing experiments)
};

// launch wrappers implemented in mwb_kernels.hip
void mwb_launch_step(const MwbDev &d, const int32_t *actions, const uint8_t *skip_mask, hipStream_t s);   // entity tasks: step_ents_kernel
void mwb_launch_clear_list(const MwbDev &d, hipStream_t s);
void mwb_launch_order(const MwbDev &d, hipStream_t s);   // envs by decreasing measured frame cost -> the map not in use
void mwb_launch_mark_reset(const MwbDev &d, const uint8_t *mask, hipStream_t s);
void mwb_view_tile(int *w, int *h);   // the tile mwb_render_view (and large observations) are rendered in
void mwb_launch_reset(const MwbDev &d, int max_blocks, hipStream_t s);   // grid-strides over reset_list
// mode 0: every env; 1: only envs with reset_set; 2: only envs without (lets reset overlap the bulk render)
void mwb_launch_prep(const MwbDev &d, int mode, hipStream_t s);
void mwb_launch_render(const MwbDev &d, int mode, hipStream_t s);
void mwb_launch_stack(const MwbDev &d, void *stack, int nstack, int dtype, int after_reset, hipStream_t s);
void mwb_launch_stack_slide(const MwbDev &d, void *stack, int nstack, int planes, int dtype, int pos, int from, int mode, hipStream_t s);
void mwb_launch_intersect(const MwbDev &d, int env, int ent, double x, double z, double radius, int *result_dev, hipStream_t s);
void mwb_launch_visible(const MwbDev &d, uint32_t *mask_out, hipStream_t s);   // get_visible_ents for every env
int mwb_launch_render_view(const MwbDev &d, hipStream_t s);   // the agent's view at d.W x d.H in tiles; 0 ok, -1 LDS, -2 HIP
void mwb_launch_top_view(const MwbDev &d, uint8_t *out, int W, int H, hipStream_t s);   // render_top_view for every env, [N][H][W][3]
int mwb_prepare_kernels(const MwbDev &d);   // 0 ok, -1 world too large for LDS, -2 HIP error, -3 frame too large for the pixel queue
size_t mwb_reset_lds_bytes(const MwbDev &d);
size_t mwb_render_lds_bytes(const MwbDev &d);
