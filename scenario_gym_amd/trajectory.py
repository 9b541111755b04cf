"""Trajectory: knot container with the reference's normalisation and query semantics.

Host-side mirror of scenario_gym/trajectory.py (reference v0.3.1).  Construction (`__init__`,
trajectory.py:34-96) runs once per entity at load time and stays on the host; the per-step
queries are executed on the device by the rollout engine -- `position_at_t` / `velocity_at_t`
here exist for API parity, setup code and tests, and follow the same arithmetic
(scipy interp1d(kind="linear")._call_linear restated in numpy).
"""
from typing import Optional, Tuple, Union

import numpy as np

_FIELDS = ("t", "x", "y", "z", "h", "p", "r")


def _lerp_rows(x, y, xq):
    """interp1d(x, y, axis=0, fill_value="extrapolate") at xq: clip(searchsorted_left, 1, n-1) bracket."""
    idx = np.clip(np.searchsorted(x, xq), 1, len(x) - 1)
    lo, hi = idx - 1, idx
    slope = (y[hi] - y[lo]) / (x[hi] - x[lo])[:, None]
    return slope * (xq - x[lo])[:, None] + y[lo]


def _resolve_heading(h):
    """trajectory.py:465-469: unwrap so that consecutive headings differ by less than pi."""
    deltas = np.diff(h) % (2 * np.pi)
    deltas = np.where(deltas > np.pi, deltas - 2 * np.pi, deltas)
    return np.hstack([h[0], deltas]).cumsum()


def is_stationary(data) -> bool:
    """trajectory.py:472-490."""
    return len(np.unique(np.where(np.isnan(data[:, 1:]), 0.0, data[:, 1:]), axis=0)) <= 1


class Trajectory:
    """(N, 7) fp64 knots [t, x, y, z, h, p, r], unique-sorted by t, read-only."""

    _fields = _FIELDS

    def __init__(self, data, fields: Tuple[str, ...] = _FIELDS):
        data = np.asarray(data)
        if fields is _FIELDS and data.ndim == 2 and data.shape[1] == 7 and data.dtype == np.float64:
            # the usual case (all seven columns, as the importers build them) without the per-column loop: same operations
            t = data[:, 0]
            if data.shape[0] > 1 and not (t[1:] > t[:-1]).all():
                data = data[np.unique(t, return_index=True)[1]]  # trajectory.py:60
            fin = np.isfinite(data).all(axis=0)
            if not (fin[0] and fin[1] and fin[2]):
                bad = "t" if not fin[0] else ("x" if not fin[1] else "y")
                raise ValueError(f"Invalid values found for {bad}. Values required for xyt.")
            out = data.copy()
            n = out.shape[0]
            for c in (3, 5, 6):
                if not fin[c]:
                    out[:, c] = 0.0
            if not fin[4]:
                if n == 1:
                    out[:, 4] = 0.0
                else:  # heading from the finite difference of xy, trajectory.py:69-78
                    tt, xy = out[:, 0], out[:, 1:3]
                    g = _lerp_rows(tt, xy, tt + 1e-2) - _lerp_rows(tt, xy, tt - 1e-2)
                    out[:, 4] = _resolve_heading(np.arctan2(g[:, 1], g[:, 0]))
            else:
                out[:, 4] = _resolve_heading(out[:, 4])
            self._data = out
            self._data.flags.writeable = False
            return
        fields = tuple(fields)
        if not all(f in fields for f in ("t", "x", "y")):
            raise ValueError("Trajectory cannot be created with t, x and y values.")
        if data.ndim != 2 or data.shape[1] != len(fields):
            raise ValueError(f"Invalid shape: {data.shape}. Expected: (N, {len(fields)}).")
        perm = [fields.index(f) for f in _FIELDS if f in fields]
        data = data[:, perm]
        data = data[np.unique(data[:, 0], return_index=True)[1]]  # trajectory.py:60
        n = data.shape[0]
        cols = []
        for f in _FIELDS:
            d = data[:, perm.index(fields.index(f))] if f in fields else np.zeros(n)
            if f not in fields or np.isfinite(d).sum() != n:
                if f == "h" and n == 1:
                    d = np.zeros(1)
                elif f == "h":  # heading from the finite difference of xy, trajectory.py:69-78
                    t = cols[0]
                    xy = np.array(cols[1:3]).T
                    g = _lerp_rows(t, xy, t + 1e-2) - _lerp_rows(t, xy, t - 1e-2)
                    d = _resolve_heading(np.arctan2(g[:, 1], g[:, 0]))
                elif f in ("z", "p", "r"):
                    d = np.zeros(n)
                else:
                    raise ValueError(f"Invalid values found for {f}. Values required for xyt.")
            elif f == "h":
                d = _resolve_heading(d)
            cols.append(np.asarray(d, np.float64))
        self._data = np.array(cols).T.copy()
        self._data.flags.writeable = False

    @classmethod
    def many(cls, datas):
        """[Trajectory(d) for d in datas], with the usual case -- (n, 7) float64 vertices as the importers build them, t
        strictly increasing, t / x / y / heading finite -- normalised for all trajectories of the same length at once (one
        scenario file: tens of trajectories, a few numpy calls instead of ~15 per trajectory).  Same operations in the same
        order per trajectory (np.diff, %, the select and cumsum run along the vertex axis): same bits as the constructor;
        anything else (unsorted times, missing headings, other shapes) goes through the constructor."""
        out = [None] * len(datas)
        groups = {}
        for i, d in enumerate(datas):
            if isinstance(d, np.ndarray) and d.ndim == 2 and d.shape[1] == 7 and d.dtype == np.float64 and d.shape[0] > 1:
                groups.setdefault(d.shape[0], []).append(i)
            else:
                out[i] = cls(d)
        for n, idx in groups.items():
            if len(idx) < 4:
                for i in idx:
                    out[i] = cls(datas[i])
                continue
            A = np.stack([datas[i] for i in idx])                          # [E, n, 7]
            t = A[:, :, 0]
            fin = np.isfinite(A).all(axis=1)                               # [E, 7]
            ok = (t[:, 1:] > t[:, :-1]).all(axis=1) & fin[:, 0] & fin[:, 1] & fin[:, 2] & fin[:, 4]
            B = A.copy()
            for c in (3, 5, 6):
                B[~fin[:, c], :, c] = 0.0
            h = B[:, :, 4]
            with np.errstate(invalid="ignore"):                            # (rows that are not ok are not used)
                deltas = np.diff(h, axis=1) % (2 * np.pi)
                deltas = np.where(deltas > np.pi, deltas - 2 * np.pi, deltas)
                B[:, :, 4] = np.concatenate([h[:, :1], deltas], axis=1).cumsum(axis=1)
            for k, i in enumerate(idx):
                if ok[k]:
                    tr = cls.__new__(cls)
                    tr._data = B[k].copy()
                    tr._data.flags.writeable = False
                    out[i] = tr
                else:
                    out[i] = cls(datas[i])
        return out

    @classmethod
    def many_arrays(cls, datas):
        """The normalised knot arrays of Trajectory.many(datas) without the Trajectory objects around them (the bulk ingest
        packs arrays: packing.load_and_pack): same operations, same bits."""
        out = [None] * len(datas)
        groups = {}
        for i, d in enumerate(datas):
            if isinstance(d, np.ndarray) and d.ndim == 2 and d.shape[1] == 7 and d.dtype == np.float64 and d.shape[0] > 1:
                groups.setdefault(d.shape[0], []).append(i)
            else:
                out[i] = cls(d)._data
        for n, idx in groups.items():
            if len(idx) < 4:
                for i in idx:
                    out[i] = cls(datas[i])._data
                continue
            A = np.stack([datas[i] for i in idx])
            t = A[:, :, 0]
            fin = np.isfinite(A).all(axis=1)
            ok = (t[:, 1:] > t[:, :-1]).all(axis=1) & fin[:, 0] & fin[:, 1] & fin[:, 2] & fin[:, 4]
            B = A.copy()
            for c in (3, 5, 6):
                B[~fin[:, c], :, c] = 0.0
            h = B[:, :, 4]
            with np.errstate(invalid="ignore"):
                deltas = np.diff(h, axis=1) % (2 * np.pi)
                deltas = np.where(deltas > np.pi, deltas - 2 * np.pi, deltas)
                B[:, :, 4] = np.concatenate([h[:, :1], deltas], axis=1).cumsum(axis=1)
            for k, i in enumerate(idx):
                out[i] = B[k] if ok[k] else cls(datas[i])._data
            if len(groups) == 1 and len(idx) == len(datas) and ok.all():
                return B  # every trajectory of the file has the same length: one [E, n, 7] block, rows already in entity order
        return out

    # ------------------------------------------------------------------ container API
    @property
    def data(self):
        return self._data

    def __len__(self):
        return len(self._data)

    def __getitem__(self, i):
        return self._data[i]

    t = property(lambda self: self._data[:, 0])
    x = property(lambda self: self._data[:, 1])
    y = property(lambda self: self._data[:, 2])
    z = property(lambda self: self._data[:, 3])
    h = property(lambda self: self._data[:, 4])
    p = property(lambda self: self._data[:, 5])
    r = property(lambda self: self._data[:, 6])

    @property
    def min_t(self):
        return self._data[0, 0]

    @property
    def max_t(self):
        return self._data[-1, 0]

    @property
    def s(self):
        ds = np.linalg.norm(np.diff(self._data[:, [1, 2]], axis=0), axis=1).cumsum()
        return np.hstack([[0.0], ds])

    @property
    def arclength(self):
        return self.s[-1]

    def is_stationary(self):
        return is_stationary(self._data)

    def translate(self, x):
        """trajectory.py:287-306: new trajectory with `x` (scalar, (7,) or (n, 7)) added to the data."""
        return self.__class__(self.data + x)

    def copy(self):
        return self.__class__(self._data.copy())

    __copy__ = copy

    def to_json(self):
        return self._data.tolist()

    # ------------------------------------------------------------------ queries (trajectory.py:142-273)
    def _interp(self, t):
        data = self._data
        if data.shape[0] == 1:  # trajectory.py:175-177
            data = np.repeat(data, 2, axis=0)
            data[-1, 0] += 1e-3
        return _lerp_rows(data[:, 0], data[:, 1:], np.atleast_1d(t))

    def position_at_t(self, t, extrapolate: Union[bool, Tuple[bool, bool]] = (False, False)) -> Optional[np.ndarray]:
        t = np.array(t, dtype=np.float64)
        if isinstance(extrapolate, tuple):
            ext_bck, ext_fwd = extrapolate
            extrapolate = True
        else:
            ext_bck = ext_fwd = extrapolate
        if t.ndim == 0:
            if not extrapolate and (t < self.min_t or t > self.max_t):
                return None
            if t < self.min_t and not ext_bck:
                return self._data[0, 1:]
            if t > self.max_t and not ext_fwd:
                return self._data[-1, 1:]
            return self._interp(t)[0]
        poses = self._interp(t)
        if not ext_bck:
            poses = np.where(t[:, None] < self.min_t, self._data[0, None, 1:], poses)
        if not ext_fwd:
            poses = np.where(t[:, None] > self.max_t, self._data[-1, None, 1:], poses)
        return poses

    def velocity_at_t(self, t, eps: float = 1e-4):
        t = np.array(t, dtype=np.float64)
        inside = np.logical_and(self.min_t <= t, t <= self.max_t)
        v_in = (self.position_at_t(t + eps / 2, extrapolate=True) - self.position_at_t(t - eps / 2, extrapolate=True)) / eps
        if t.ndim >= 1:
            inside = inside.reshape(-1, 1)
        return np.where(inside, v_in, np.zeros(t.shape + (6,)))
