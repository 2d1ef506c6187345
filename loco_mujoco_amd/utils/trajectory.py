"""
Trajectory handling for ``reset()`` and ``create_dataset()``.

Same surface and semantics as the reference's ``loco_mujoco/utils/trajectory.py`` (constructor
``:17-102``, ``create_dataset`` ``:104-151``, cubic re-sampling to the control rate ``:184-234``,
``reset_trajectory`` ``:236-273`` including its ``np.random`` draw order ``:253,:259`` and the
"re-centre x/y on the sampled step" rule ``:268-269``); storage is one contiguous array per
observation key so that the batched front-end can export a flat table of initial states for the
device-side auto-reset (``Trajectory.as_state_table``).
"""

import warnings

import numpy as np
from scipy import interpolate


class Trajectory:

    def __init__(self, keys, low, high, joint_pos_idx, interpolate_map, interpolate_remap,
                 traj_path=None, traj_files=None, interpolate_map_params=None, interpolate_remap_params=None,
                 traj_dt=0.002, control_dt=0.01, ignore_keys=None, clip_trajectory_to_joint_ranges=False,
                 traj_info=None, warn=True):
        if (traj_path is None) == (traj_files is None):
            raise AssertionError("Please specify either traj_path or traj_files, but not both.")

        if traj_path is not None:
            with np.load(traj_path, allow_pickle=True) as f:
                self._trajectory_files = {k: f[k] for k in f.files}
        else:
            self._trajectory_files = {k: np.asarray(v) for k, v in traj_files.items()}

        self.check_if_trajectory_is_in_range(low, high, keys, joint_pos_idx, warn, clip_trajectory_to_joint_ranges)

        # goals live only in the trajectory file (keys starting with "goal"), append them
        keys = list(keys)
        keys += [k for k in self._trajectory_files if k.startswith("goal") and k not in keys]
        for ik in (ignore_keys or []):
            keys.remove(ik)
        self.keys = keys

        if "split_points" in self._trajectory_files:
            self.split_points = np.asarray(self._trajectory_files["split_points"])
        else:
            self.split_points = np.array([0, len(next(iter(self._trajectory_files.values())))])

        self.trajectories = self._extract_trajectory_from_files()

        if traj_info is not None:
            assert len(traj_info) == self.number_of_trajectories, \
                "The number of trajectory infos/labels need to be equal to the number of trajectories."
        self._traj_info = traj_info

        self.traj_dt = traj_dt
        self.control_dt = control_dt
        if self.traj_dt != control_dt:
            self._interpolate_trajectories(interpolate_map, interpolate_remap,
                                           interpolate_map_params, interpolate_remap_params)

        self.subtraj_step_no = 0
        self.traj_no = 0
        self.subtraj = self._get_subtraj(self.traj_no)

    # ------------------------------------------------------------------ construction helpers
    def _extract_trajectory_from_files(self):
        cols = [np.asarray(self._trajectory_files[k]) for k in self.keys]
        n = len(cols[0])
        assert all(len(c) == n for c in cols), "Some observations have different lengths than others. " \
                                              "Trajectory is corrupted. "
        out = []
        for c in cols:
            parts = np.split(c, self.split_points[1:-1])
            assert all(len(p) == len(parts[0]) for p in parts), "Only trajectories of equal length are currently " \
                                                              "supported."
            out.append(np.array(parts))
        return out

    def _interpolate_trajectories(self, map_funct, re_map_funct, map_params, re_map_params):
        assert (map_funct is None) == (re_map_funct is None)
        n_old = self.trajectory_length
        x = np.arange(n_old)
        n_new = round(n_old * (self.traj_dt / self.control_dt))
        x_new = np.linspace(0, n_old - 1, n_new, endpoint=True)

        per_traj = []
        for i in range(self.number_of_trajectories):
            traj = [obs[i] for obs in self.trajectories]
            if map_funct is not None:
                traj = map_funct(traj) if map_params is None else map_funct(traj, **map_params)
            new_traj = interpolate.interp1d(x, traj, kind="cubic", axis=1)(x_new)
            if re_map_funct is not None:
                new_traj = re_map_funct(new_traj) if re_map_params is None else re_map_funct(new_traj, **re_map_params)
            per_traj.append(new_traj)

        self.trajectories = [np.array([t[k] for t in per_traj]) for k in range(len(self.keys))]
        lens = [len(self.trajectories[0][k]) for k in range(self.number_of_trajectories)]
        self.split_points = np.concatenate([[0], np.cumsum(lens)])

    def check_if_trajectory_is_in_range(self, low, high, keys, j_idx, warn, clip_trajectory_to_joint_ranges):
        if not (warn or clip_trajectory_to_joint_ranges):
            return
        j_idx = list(j_idx[2:])                                  # x and y are not part of the obs space
        highs = dict(zip(keys[2:], high))
        lows = dict(zip(keys[2:], low))
        for i, (k, d) in enumerate(self._trajectory_files.items()):
            if i in j_idx and k in keys:
                if warn:
                    clip_message = "Clipping the trajectory into range!" if clip_trajectory_to_joint_ranges else ""
                    if np.max(d) > highs[k]:
                        warnings.warn("Trajectory violates joint range in %s. Maximum in trajectory is %f "
                                      "and maximum range is %f. %s" % (k, np.max(d), highs[k], clip_message),
                                      RuntimeWarning)
                    elif np.min(d) < lows[k]:
                        warnings.warn("Trajectory violates joint range in %s. Minimum in trajectory is %f "
                                      "and minimum range is %f. %s" % (k, np.min(d), lows[k], clip_message),
                                      RuntimeWarning)
                if clip_trajectory_to_joint_ranges:
                    self._trajectory_files[k] = np.clip(d, lows[k], highs[k])

    # ------------------------------------------------------------------ datasets
    def create_dataset(self, ignore_keys=None, state_callback=None, state_callback_params=None):
        flat = dict(zip(self.keys, [a.copy() for a in self.flattened_trajectories()]))
        for ik in (ignore_keys or []):
            del flat[ik]
        states = np.concatenate(list(flat.values()), axis=1)
        if state_callback is not None:
            states = np.array([state_callback(s, **(state_callback_params or {})) for s in states])

        pieces = np.split(states, self.split_points[1:-1])
        cur = np.concatenate([p[:-1] for p in pieces])
        nxt = np.concatenate([p[1:] for p in pieces])
        absorbing = np.zeros(len(cur))
        last = np.concatenate([np.concatenate([np.zeros(len(p) - 2), [1.0]]) for p in pieces])
        out = dict(states=cur, next_states=nxt, absorbing=absorbing, last=last)
        if self._traj_info is not None:
            out["info"] = np.array([[lab] * self.trajectory_length for lab in self._traj_info]).reshape(-1)
        return out

    # ------------------------------------------------------------------ sampling
    def reset_trajectory(self, substep_no=None, traj_no=None):
        if traj_no is None:
            self.traj_no = np.random.randint(0, self.number_of_trajectories)
        else:
            assert 0 <= traj_no <= self.number_of_trajectories
            self.traj_no = traj_no
        if substep_no is None:
            self.subtraj_step_no = np.random.randint(0, self.trajectory_length)
        else:
            assert 0 <= substep_no <= self.trajectory_length
            self.subtraj_step_no = substep_no

        self.subtraj = self._get_subtraj(self.traj_no)
        # x and y are made relative to the sampled step
        self.subtraj[0] -= self.subtraj[0][self.subtraj_step_no]
        self.subtraj[1] -= self.subtraj[1][self.subtraj_step_no]
        return [obs[self.subtraj_step_no] for obs in self.subtraj]

    def get_current_sample(self):
        return self._get_ith_sample_from_subtraj(self.subtraj_step_no)

    def get_next_sample(self):
        self.subtraj_step_no += 1
        if self.subtraj_step_no == self.trajectory_length:
            return None
        return self._get_ith_sample_from_subtraj(self.subtraj_step_no)

    def get_from_sample(self, sample, key):
        assert len(sample) == len(self.keys)
        return sample[self.get_idx(key)]

    def get_idx(self, key):
        return self.keys.index(key)

    def flattened_trajectories(self):
        out = []
        for obs in self.trajectories:
            if obs.ndim == 2:
                out.append(obs.reshape((-1, 1)))
            elif obs.ndim == 3:
                out.append(obs.reshape((-1, obs.shape[2])))
            else:
                raise ValueError("Unsupported shape of observation %s." % (obs.shape,))
        return out

    def as_state_table(self):
        """
        All samples as one float64 array ``(n_traj*n_samples, sum of key dims)`` in key order, with x/y
        (first two keys) zeroed — every row is what ``reset_trajectory`` would return for that (traj, step).
        Used to fill the device-side reset table.
        """
        tab = np.concatenate(self.flattened_trajectories(), axis=1)
        tab[:, 0:2] = 0.0
        return tab

    def _get_subtraj(self, i):
        return [obs[i].copy() for obs in self.trajectories]

    def _get_ith_sample_from_subtraj(self, i):
        return [np.array(obs[i].copy()).flatten() for obs in self.subtraj]

    @property
    def number_obs_trajectory(self):
        return len(self.trajectories)

    @property
    def trajectory_length(self):
        return self.trajectories[0].shape[1]

    @property
    def number_of_trajectories(self):
        return self.trajectories[0].shape[0]
