"""Recorded-pedestrian data either side of the rollout path:

  * TrajNet++ ndjson ingest -> per-scene [T, P, 5] observation tensors (px, py, vx, vy, radius) + presence masks,
    what the reference builds frame by frame as lists of ObservableState
    (crowd_nav/utils/misc.py:47-187 `GetRealData`, `Convert_to_ObserState`, `GetState`, `GetIndex`, `GetVel`;
    scene joining / windowing of the vendored reader, trajnetplusplustools/reader.py:44-93);
  * the raw_memory rows DataGen replays (misc.py:85-89) and the (state, next velocity) pairs the world models are
    trained on (misc.py:118-123 `StoreAction`);
  * the SGAN text cache, one `frame<TAB>ped<TAB>x<TAB>y` line per observation
    (misc.py:92-111, explorer.py:116-121, reader: sgan/sdata/trajectories.py:39-50).

Host code (numpy): this is file parsing and a few thousand rows per dataset, done once; the tensors it returns are
what `VecDataGen` / the world-model trainers move to HBM.
"""
import json
import os
from collections import OrderedDict, defaultdict

import numpy as np


# ---------------------------------------------------------------------------------------------- ndjson
def read_ndjson(path):
    """reader.py:29-43.  Returns (tracks_by_frame: frame -> [(frame, ped, x, y), ...] in file order,
    scenes_by_id: id -> dict(id, ped, start, end, fps, tag) in file order)."""
    tracks = defaultdict(list)
    scenes = OrderedDict()
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            row = json.loads(line)
            t = row.get("track")
            if t is not None:
                tracks[t["f"]].append((t["f"], t["p"], t["x"], t["y"]))
                continue
            s = row.get("scene")
            if s is not None:
                scenes[s["id"]] = dict(id=s["id"], ped=s["p"], start=s["s"], end=s["e"], fps=s.get("fps"), tag=s.get("tag"))
    return tracks, scenes


def _frames_with_tracks(tracks, start, end):
    return sorted({f for f in range(start, end + 1) if tracks.get(f)})


def join_scenes(tracks, scenes, stride=-1, windows_size=-1):
    """reader.py:44-93 (`joinScene` / `joinDuration`): overlapping scenes (sorted by start) are merged into
    continuous durations; with stride > 0 and windows_size > 0 every duration is cut into windows of
    `windows_size + 1` recorded frames, `stride` frames apart.  Returns (joined scenes: OrderedDict id -> dict,
    full durations [[start, end], ...])."""
    ids = sorted(scenes, key=lambda k: scenes[k]["start"])
    durations = [[scenes[i]["start"], scenes[i]["end"]] for i in ids]
    j_dur, s_id = [list(durations[0])], [ids[0]]
    for i, d in enumerate(durations[1:]):
        if j_dur[-1][0] <= d[0] <= j_dur[-1][1]:
            j_dur[-1][1] = d[1]                      # (the reference does not take the max of the two ends)
        else:
            j_dur.append(list(d))
            s_id.append(ids[i + 1])
    full = j_dur
    if stride > 0 and windows_size > 0:
        w_dur, w_id = [], []
        for i, d in enumerate(j_dur):
            fids = _frames_with_tracks(tracks, d[0], d[1])
            for j in range(0, len(fids) + 1, stride):
                if j + windows_size > len(fids) - 1:
                    break
                w_dur.append([fids[j], fids[j + windows_size]])
                w_id.append(s_id[i])
        j_dur, s_id = w_dur, w_id
    joined = OrderedDict()
    for i, dur in enumerate(j_dur):
        src = scenes[s_id[i]]
        joined[i] = dict(id=i, ped=tracks[dur[0]][0][1], start=dur[0], end=dur[1], fps=src["fps"], tag=src["tag"])
    return joined, full


def scene_paths(tracks, scene):
    """reader.py:109-120,147-166: rows of the scene's frames grouped per pedestrian, primary pedestrian first, the
    others in order of first appearance.  Returns (rows, paths) with paths = list of [(frame, ped, x, y), ...]."""
    rows = [r for f in range(scene["start"], scene["end"] + 1) for r in tracks.get(f, [])]
    primary, others = [], OrderedDict()
    for r in rows:
        if r[1] == scene["ped"]:
            primary.append(r)
        else:
            others.setdefault(r[1], []).append(r)
    return rows, [primary] + list(others.values())


# ---------------------------------------------------------------------------------------------- observations
def scene_tensor(fps, paths, frame_ids, radius=0.3, padding_last="stay", padding_first="none"):
    """misc.py:126-187 for a whole scene at once.  Returns (obs [T,P,5] float64, present [T,P] bool): the
    reference's per-frame list of ObservableState is obs[t][present[t]] (pedestrians that have not appeared yet are
    skipped when padding_first == 'none'; with 'stay' they wait at their first position and every row is present)."""
    T, P = len(frame_ids), len(paths)
    fids = np.asarray(frame_ids)
    obs = np.zeros((T, P, 5), np.float64)
    present = np.zeros((T, P), bool)
    obs[..., 4] = radius
    for p, path in enumerate(paths):
        if not path:
            continue
        pf = np.array([r[0] for r in path])
        xy = np.array([[r[2], r[3]] for r in path], np.float64)
        L = len(path)
        # GetIndex: -1 before the first frame or inside a gap; a "fake" index >= L after the last frame
        idx = np.full(T, -1, np.int64)
        first_at = {}
        for i, f in enumerate(pf.tolist()):
            first_at.setdefault(f, i)                                         # first match wins, like the loop
        for t, c in enumerate(fids.tolist()):
            if c < pf[0]:
                continue
            if c > pf[-1]:
                idx[t] = L + int(np.count_nonzero((fids > pf[-1]) & (fids <= c)))
            else:
                idx[t] = first_at.get(c, -1)
        if padding_first == "stay":
            idx[idx == -1] = 0
        here = idx >= 0
        present[:, p] = here
        # GetVel: zero at the first sample and after the track ended
        vel = np.zeros((L, 2), np.float64)
        vel[1:] = (xy[1:] - xy[:-1]) * fps
        for t in np.nonzero(here)[0]:
            i = int(idx[t])
            if i >= L:
                if padding_last == "moving":
                    lv = vel[L - 1]
                    obs[t, p, 0] = xy[-1, 0] + (lv[0] / fps) * (i - L)
                    obs[t, p, 1] = xy[-1, 1] + (lv[1] / fps) * (i - L)
                    obs[t, p, 2:4] = lv
                else:                                   # 'stay': waits at its last position
                    obs[t, p, 0:2] = xy[-1]
            else:
                obs[t, p, 0:2] = xy[i]
                obs[t, p, 2:4] = vel[i]
    return obs, present


class RealData(object):
    """What GetRealData returns, as arrays.  scenes: list of dict(id, fps, obs [T,P,5], present [T,P],
    start_ends [P,4], frame_ids)."""

    def __init__(self, scenes, full_durations, cache_rows):
        self.scenes, self.full_durations, self.cache_rows = scenes, full_durations, cache_rows

    def raw_memory(self, as_states=False):
        """misc.py:85-89 rows `(ob, 0, done, Nothing(), start_ends)`; ob = [n,5] array per frame (or the
        reference's list[ObservableState] with as_states=True)."""
        from ..envs.utils.info import Nothing
        from ..envs.utils.state import ObservableState
        rows = []
        for sc in self.scenes:
            T = sc["obs"].shape[0]
            se = sc["start_ends"].tolist()
            for t in range(T):
                ob = sc["obs"][t][sc["present"][t]]
                if as_states:
                    ob = [ObservableState(*r) for r in ob.tolist()]
                rows.append((ob, 0, t == T - 1, Nothing(), se))
        return rows

    def world_pairs(self):
        """misc.py:118-123 `StoreAction(memory, cur_obs, last_obs)` for every consecutive frame pair of every
        scene: (current_s [n,4] = px,py,vx,vy of the earlier frame, next_action [n,2] = velocities of the later
        frame, cut to the earlier frame's pedestrian count)."""
        out = []
        for sc in self.scenes:
            for t in range(1, sc["obs"].shape[0]):
                last = sc["obs"][t - 1][sc["present"][t - 1]]
                cur = sc["obs"][t][sc["present"][t]]
                out.append((last[:, 0:4].astype(np.float32), cur[:len(last), 2:4].astype(np.float32)))
        return out

    def episode_tensor(self):
        """Scenes whose pedestrian count never changes, stacked: (obs [n,T_max,N,5], lengths [n], scene ids) -- the
        layout VecDataGen keeps in HBM.  Raises if no scene qualifies."""
        ok = [sc for sc in self.scenes if sc["present"].all()]
        if not ok:
            raise ValueError("no scene with a constant pedestrian count (use padding_first='stay')")
        N = max(set(sc["obs"].shape[1] for sc in ok), key=[sc["obs"].shape[1] for sc in ok].count)
        ok = [sc for sc in ok if sc["obs"].shape[1] == N]
        T = max(sc["obs"].shape[0] for sc in ok)
        obs = np.zeros((len(ok), T, N, 5), np.float64)
        for i, sc in enumerate(ok):
            obs[i, :sc["obs"].shape[0]] = sc["obs"]
        return obs, np.array([sc["obs"].shape[0] for sc in ok]), [sc["id"] for sc in ok]


def get_real_data(dataset_file, phase="train", stride=-1, windows_size=-1, padding_last="stay", padding_first="none",
                  dataset_slice=None, cache_dir=None, radius=0.3):
    """misc.py:47-116 `GetRealData` without the ReplayMemory containers.  The scene selection reproduces the
    reference's arithmetic, quirks included (`limit`/`start` only take effect when limit > 0, reader.py:100-101)."""
    tracks, scenes = read_ndjson(dataset_file)
    joined, full = join_scenes(tracks, scenes, stride, windows_size)
    limit, start, total = -1, 0, len(joined)
    if dataset_slice is not None:
        start, total = dataset_slice[0], dataset_slice[1]
        limit = total
    if phase == "train":
        limit = int(0.7 * total)
    if phase == "val":
        start = int(0.7 * total)
        limit = total - start
    ids = list(joined.keys())
    if limit > 0:
        ids = ids[start:start + limit]
    out, cache_rows = [], []
    for sid in ids:
        sc = joined[sid]
        rows, paths = scene_paths(tracks, sc)
        fids = _frames_with_tracks(tracks, sc["start"], sc["end"])
        obs, present = scene_tensor(sc["fps"], paths, fids, radius, padding_last, padding_first)
        start_ends = np.array([[p[0][2], p[0][3], p[-1][2], p[-1][3]] for p in paths], np.float64)
        out.append(dict(id=sid, fps=sc["fps"], obs=obs, present=present, start_ends=start_ends, frame_ids=fids))
        if cache_dir is not None:
            cache_rows += [[r[0], r[1], r[2], r[3]] for r in rows]
    data = RealData(out, full, cache_rows)
    if cache_dir is not None:
        write_scene_caches(cache_dir, cache_rows, full)
    return data


# ---------------------------------------------------------------------------------------------- SGAN text cache
def write_sgan_cache(path, rows, mode="w"):
    """One `frame<TAB>ped<TAB>x<TAB>y` line per row, values printed with str() as the reference's `"%s"` does
    (misc.py:110-111, explorer.py:118-121, datagen.py:424-430, world_model.py:238-240)."""
    with open(path, mode) as f:
        for r in rows:
            f.write("%s\t%s\t%s\t%s\n" % (r[0], r[1], r[2], r[3]))


def read_sgan_cache(path, delim="\t"):
    """sgan/sdata/trajectories.py:39-50 `read_file`: float rows [frame, ped, x, y]."""
    if delim == "tab":
        delim = "\t"
    elif delim == "space":
        delim = " "
    data = []
    with open(path, "r") as f:
        for line in f:
            data.append([float(v) for v in line.strip().split(delim)])
    return np.asarray(data)


def write_scene_caches(cache_dir, cache_rows, full_durations):
    """misc.py:101-111: the unique rows of all loaded scenes, one numbered file per continuous duration that has
    rows.  Returns the file paths."""
    if not cache_rows:
        return []
    rows = np.unique(np.asarray(cache_rows, np.float64), axis=0)
    paths, fcount = [], 0
    for d in full_durations:
        sel = rows[(rows[:, 0] >= d[0]) & (rows[:, 0] <= d[1])]
        if len(sel) > 0:
            fcount += 1
            p = os.path.join(cache_dir, "%d.txt" % fcount)
            write_sgan_cache(p, sel)
            paths.append(p)
    return paths


def history_from_cache(rows, obs_len=8):
    """The [obs_len, N, 2] position history SGANWorld.data_loader builds from a cache file
    (world_model.py:152-232): last obs_len frames, pedestrians in id order, positions rounded to 1e-4, a track that
    starts late / ends early padded with its first / last position.  Feeds VecSGANWorld.reset_history."""
    rows = np.asarray(rows, np.float64)
    frames = np.unique(rows[:, 0]).tolist()[-obs_len:]
    rows = rows[np.isin(rows[:, 0], frames)]
    peds = np.unique(rows[:, 1]).tolist()
    hist = np.zeros((len(frames), len(peds), 2), np.float64)
    for j, pid in enumerate(peds):
        pr = np.around(rows[rows[:, 1] == pid], decimals=4)
        pr = pr[np.argsort(pr[:, 0], kind="stable")]
        for i, f in enumerate(frames):
            seen = pr[pr[:, 0] <= f]
            hist[i, j] = seen[-1, 2:4] if len(seen) else pr[0, 2:4]      # a gap inside a track repeats its last sample
    return hist, peds
