"""Single-environment Gym-style view (a batch of one) for code written against
gym_miniworld.miniworld.MiniWorldEnv: seed / reset / step / render_obs / render_depth and the
attributes callers and the reference's run_tests.py read (miniworld.py:425-531, 658-716,
933-971, 1160-1220).  Plain Gym semantics: step() returns the terminal observation and the
caller resets.
"""
from enum import IntEnum

import numpy as np

from .vec_env import Box, Discrete


class _Ent:
    def __init__(self, radius, height):
        self.pos, self.dir, self.radius, self.height = None, None, radius, height

    @property
    def dir_vec(self):   # entity.py:72-80
        return np.array([np.cos(self.dir), 0, -np.sin(self.dir)])

    @property
    def right_vec(self):   # entity.py:82-90
        return np.array([np.sin(self.dir), 0, np.cos(self.dir)])


COLOR_NAMES = ("blue", "green", "grey", "purple", "red", "yellow")   # sorted(COLORS), entity.py:18
_KIND_NAMES = ("Box", "MeshEnt", "ImageFrame", "TextFrame")


class _Room:
    def __init__(self, words):
        if len(words) == 24:
            self.min_x, self.max_x, self.min_z, self.max_z, self.wall_height = (float(w) for w in words[:5])
            self.outline = np.array([[self.max_x, 0, self.max_z], [self.max_x, 0, self.min_z], [self.min_x, 0, self.min_z], [self.min_x, 0, self.max_z]])
        else:   # polygon room table (YMaze): height, tex, n_edges | culled << 8, pad, then 4 edges of 12 words starting p.x p.z
            ne = int(np.asarray(words[2:3]).view(np.int32)[0]) & 255
            self.wall_height = abs(float(words[0]))
            self.outline = np.array([[float(words[4 + 12 * k]), 0.0, float(words[5 + 12 * k])] for k in range(ne)])
            self.min_x, self.max_x = self.outline[:, 0].min(), self.outline[:, 0].max()
            self.min_z, self.max_z = self.outline[:, 2].min(), self.outline[:, 2].max()
        self.num_walls = len(self.outline)


class MiniWorldEnv:
    class Actions(IntEnum):   # miniworld.py:437-454
        turn_left = 0
        turn_right = 1
        move_forward = 2
        move_back = 3
        pickup = 4
        drop = 5
        toggle = 6
        done = 7

    def __init__(self, env_id="MiniWorld-OneRoom-v0", domain_rand=False, obs_width=80, obs_height=60, device=0,
                 seed=None, **kwargs):
        from .batch import BatchedMiniWorld
        self._b = BatchedMiniWorld(env_id, num_envs=1, domain_rand=domain_rand, obs_width=obs_width,
                                   obs_height=obs_height, want_depth=True, layout="HWC", device=device,
                                   auto_reset=False, **kwargs)
        self.actions = MiniWorldEnv.Actions
        self.action_space = Discrete(self._b.n_actions)   # e.g. envs/hallway.py:23; simtorealpush.py:37 adds move_back
        self.observation_space = Box(0, 255, (obs_height, obs_width, 3), np.uint8)
        self.max_episode_steps = self._b.max_episode_steps
        self.params = self._b.params
        self.agent, self.box = _Ent(self._b.agent_radius, 1.6), _Ent(float(np.sqrt(0.8 * 0.8 + 0.8 * 0.8) / 2), 0.8)
        self.agent.carrying = None
        self.entities = [self.box, self.agent]
        if self._b.n_boxes == 2 and not self._b.ent_task:   # the two-box T-maze, tmaze.py:166-169
            self.red_box, self.blue_box = self.box, _Ent(self.box.radius, 0.8)
            self.boxes = [self.red_box, self.blue_box]
            self.entities = [self.red_box, self.blue_box, self.agent]
            self.goal_box_idx, self.penalty_box_idx = 0, 1
            if self._b.task == "SimToRealPush":   # simtorealpush.py:88-99
                self.box1, self.box2 = self.red_box, self.blue_box
        elif self._b.n_boxes > 2 and not self._b.ent_task:   # PutNext: one box per colour in COLOR_NAMES order (putnext.py:31-38, entity.py:18)
            self.boxes = [self.box] + [_Ent(self.box.radius, 0.8) for _ in range(self._b.n_boxes - 1)]
            for b, name in zip(self.boxes, ("blue", "green", "grey", "purple", "red", "yellow")):
                b.color = name
            self.red_box, self.yellow_box = self.boxes[4], self.boxes[5]
            self.entities = self.boxes + [self.agent]
        if self._b.ent_task:   # a general entity list: rebuilt from the device state after every reset / step
            self.entities, self._slots = [self.agent], []
            del self.box
        self.rooms, self.step_count = [], 0
        self._seeded = False
        if seed is not None:
            self.seed(seed)

    @property
    def domain_rand(self):
        return bool(self._b.domain_rand)

    @domain_rand.setter
    def domain_rand(self, flag):   # run_tests.py:64-66 toggles it after construction
        self._b.set_domain_rand(flag)

    def seed(self, seed=None):
        if seed is None:   # the reference seeds from entropy (miniworld.py:522,528-530)
            seed = int(np.random.SeedSequence().entropy % (2 ** 63))
        self._b.seed(np.array([seed], dtype=np.uint64))
        self._seeded = True
        return [seed]

    def _sync_entities(self, st, rebuild):
        """the tasks with mesh entities / frames: one object per slot (kind, mesh_name, color, size ..., entity.py), self.entities in
        the reference's current LIST order (PickupObjs removes entries, CollectHealth moves a respawned kit to the end)"""
        from . import _lib
        B = self._b.n_boxes
        if rebuild or len(self._slots) != B:
            self._slots = []
            for k in range(B):
                ent = _Ent(float(st["ent_radius"][0, k]), float(st["ent_height"][0, k]))
                kind = int(st["ent_kind"][0, k])
                ent.kind = _KIND_NAMES[kind]
                ent.is_static = bool(st["ent_static"][0, k])
                ci = int(st["ent_color"][0, k])
                if kind == 0:
                    s0 = float(st["boxes_size"][0, k])
                    ent.color, ent.size = COLOR_NAMES[ci], np.array([s0, s0, s0])
                elif kind == 1:
                    geom = _lib.MESH_GEOMS[int(st["ent_geom"][0, k])]
                    ent.mesh_name = geom + "_" + COLOR_NAMES[ci] if ci >= 0 else geom
                    ent.scale = float(st["ent_scale"][0, k])
                else:
                    ent.width, ent.depth = float(st["boxes_size"][0, k]), 0.05
                self._slots.append(ent)
        for k, ent in enumerate(self._slots):
            ent.pos, ent.dir = st["boxes_pos"][0, k].copy(), float(st["boxes_dir"][0, k])
            if ent.kind == "Box":
                ent.color_vec = st["boxes_color"][0, k].copy()
        self.entities = [self.agent if v == -2 else self._slots[v] for v in st["ent_order"][0] if v != -1]
        c = int(st["carrying"][0])
        self.agent.carrying = None if c < 0 else self._slots[c]
        task = self._b.task
        if task in ("Sidewalk", "WallGap"):
            self.box = self._slots[6 if task == "Sidewalk" else 0]
        if task == "CollectHealth":
            self.health = float(st["task_f"][0])
        if task == "PickupObjs":
            self.num_picked_up = int(st["task_i"][0])
        if task == "Sign":   # sign.py:91-102: (boxes, keys) x (blue, red, green)
            self._objects = [tuple(self._slots[0:3]), tuple(self._slots[3:6])]

    def _sync(self, rebuild=False):
        st = self._b.get_state()
        self.agent.pos, self.agent.dir = st["agent_pos"][0].copy(), float(st["agent_dir"][0])
        if self._b.ent_task:
            self._sync_entities(st, rebuild)
            (self.agent.cam_height, self.agent.cam_fwd_disp, self.agent.cam_pitch, self.agent.cam_fov_y) = st["cam"][0]
            self.sky_color, self.light_pos = st["sky_color"][0], st["light_pos"][0]
            self.light_color, self.light_ambient = st["light_color"][0], st["light_ambient"][0]
            self.step_count = int(st["step_count"][0])
            return
        self.box.pos, self.box.dir = st["box_pos"][0].copy(), float(st["box_dir"][0])
        self.box.color_vec = st["box_color"][0].copy()
        s0 = float(st["box_size"][0])
        self.box.radius, self.box.height, self.box.size = float(np.sqrt(s0 * s0 + s0 * s0) / 2), s0, np.array([s0, s0, s0])
        if self._b.n_boxes == 2:
            self.blue_box.pos, self.blue_box.dir = st["box2_pos"][0].copy(), float(st["box2_dir"][0])
            self.blue_box.color_vec = st["box2_color"][0].copy()
            s1 = float(st["box2_size"][0])
            self.blue_box.radius, self.blue_box.height = float(np.sqrt(s1 * s1 + s1 * s1) / 2), s1
            self.blue_box.size = np.array([s1, s1, s1])
            self.goal_dist = float(st["goal_dist"][0])
            self.goal_box_idx = int(st["goal_idx"][0])
            self.penalty_box_idx = 1 - self.goal_box_idx
        if self._b.n_boxes > 2:
            for k, b in enumerate(self.boxes):
                b.pos, b.dir, b.color_vec = st["boxes_pos"][0, k].copy(), float(st["boxes_dir"][0, k]), st["boxes_color"][0, k].copy()
                sk = float(st["boxes_size"][0, k])
                b.radius, b.height, b.size = float(np.sqrt(sk * sk + sk * sk) / 2), sk, np.array([sk, sk, sk])
        c = int(st["carrying"][0])
        self.agent.carrying = None if c < 0 else (self.boxes[c] if hasattr(self, "boxes") else self.box)
        (self.agent.cam_height, self.agent.cam_fwd_disp, self.agent.cam_pitch, self.agent.cam_fov_y) = st["cam"][0]
        self.sky_color, self.light_pos = st["sky_color"][0], st["light_pos"][0]
        self.light_color, self.light_ambient = st["light_color"][0], st["light_ambient"][0]
        self.step_count = int(st["step_count"][0])

    def reset(self):
        if not self._seeded:
            self.seed()
        obs = self._b.reset().cpu().numpy()[0]
        rooms, _ = self._b.get_geometry(0)
        self.rooms = [_Room(w) for w in rooms]
        self._sync(rebuild=True)
        if self._b.task == "Sign":   # sign.py:130-133: the observation is a dict
            return {"obs": obs, "goal": int(self._b.task_args[2])}
        return obs

    def step(self, action):
        import torch
        self._b.step(torch.tensor([int(action)], dtype=torch.int32))
        obs = self._b.obs.cpu().numpy()[0]
        reward, done = float(self._b.reward64.cpu()[0]), bool(self._b.done.cpu()[0])
        self._sync()
        info = {}
        if self._b.has_goal_pos:   # tmaze.py:66,206
            info["goal_pos"] = self._b.goal_pos.cpu().numpy()[0].copy()
        if self._b.has_features:   # tmaze.py:311-318
            info["feature"] = self._b.feature.cpu().numpy()[0].astype(np.float64)
        if self._b.has_health:   # collecthealth.py:75
            info["health"] = float(self._b.feature.cpu().numpy()[0, 0])
        if self._b.task == "Sign":
            obs = {"obs": obs, "goal": int(self._b.task_args[2])}
        return obs, reward, done, info

    def render_obs(self):
        return self._b.render().cpu().numpy()[0]

    def render_depth(self):
        self._b.render()
        return self._b.depth.cpu().numpy()[0]

    def get_visible_ents(self):
        """miniworld.py:1222-1315: the set of entities (never the agent) whose query cube is visible"""
        mask = int(self._b.visible_ents().cpu().numpy()[0])
        if self._b.ent_task:   # bits name slots; entities that left the list have no cube
            return {ent for i, ent in enumerate(self._slots) if (mask >> i) & 1 and any(ent is x for x in self.entities)}
        return {ent for i, ent in enumerate(self.entities[:-1]) if (mask >> i) & 1}

    def render_top_view(self, width=None, height=None):
        """miniworld.py:1087-1158 (frame size: the reference passes a frame buffer; default the observation's)"""
        return self._b.render_top_view(width, height).cpu().numpy()[0]

    def render(self, mode="rgb_array", close=False, view="agent"):
        """miniworld.py:1317-1335, mode 'rgb_array' only (there is no window here): the 800 x 600 human-view frame"""
        if close:
            return None
        assert view in ("agent", "top") and mode == "rgb_array", "only mode='rgb_array' exists without a window"
        if view == "top":
            return self.render_top_view(800, 600)
        return self._b.render_view(800, 600).cpu().numpy()[0]   # render_obs(self.vis_fb): window_width x window_height (miniworld.py:461-462,505)

    def intersect(self, ent, pos, radius):
        """miniworld.py:933-959: True for a wall, the other entity for an entity hit, else None."""
        if self._b.ent_task:   # results name slots (2 + slot; 2 + n_slots = the agent)
            idx = self._b.n_boxes if ent is self.agent else next((i for i, e2 in enumerate(self._slots) if e2 is ent), -1)
            r = self._b.intersect(0, pos[0], pos[2], radius, ent=idx)
            return True if r == 1 else None if r < 2 else (self.agent if r - 2 == self._b.n_boxes else self._slots[r - 2])
        idx = next((i for i, e2 in enumerate(self.entities) if e2 is ent), -1)
        r = self._b.intersect(0, pos[0], pos[2], radius, ent=idx)
        if r == 1:
            return True
        return self.entities[r - 2] if r >= 2 else None

    def near(self, ent0, ent1=None):   # miniworld.py:961-971
        ent1 = ent1 or self.agent
        dist = np.linalg.norm(ent0.pos - ent1.pos)
        return dist < ent0.radius + ent1.radius + 1.1 * self.params.get_max("forward_step")

    def close(self):
        self._b.close()


def make(env_id, **kwargs):
    """gym.make('MiniWorld-<Class>-v0') for the covered ids (envs/__init__.py:43-49)."""
    return MiniWorldEnv(env_id, **kwargs)
