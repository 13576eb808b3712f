"""SECOND, INDEPENDENT restatement of SPEC.md -- TEST INFRASTRUCTURE ONLY, never imported by the product.

Why it exists ("What's weak" 1): parity is pinned to oracle/oracle.c, and oracle.c is pinned only by the builder's own
known-answer tests and takes its constraint ORDER from the product's planner (sb_plan_get_order). A misreading of SPEC.md shared by
oracle.c and the kernels -- both were written one operation at a time against each other -- would pass every test. This file is
written from SPEC.md alone, deliberately differently:

  * pure numpy, binary32 throughout, one correctly rounded IEEE operation per numpy ufunc call (numpy never contracts a*b+c and
    never reassociates element-wise arithmetic), component arrays instead of xyz triples;
  * no sequential walk: the constraints are grouped by an edge colouring made HERE (greedy, per constraint type, natural order --
    `greedy_colour_order`), and a colour class is projected as ONE vectorised update, which equals any sequential order of that
    class because its constraints share no particle;
  * its own order generator: the flat order it publishes (colour by colour) is fed to the C oracle, so the planner of the plugin
    is not involved anywhere in tests/test_independent_oracle.py.

The reference tree holds nothing to restate (/root/reference/README.md:1 is its only line): PARITY UNPINNED stays true; what this
file adds is independence between the two checkers.
"""
import numpy as np

F = np.float32
_TINY = F(2.0 ** -96)


def greedy_colour_order(n_particles, dist_ij, vol_ijkl, bend_ijkl):
    """Greedy colouring per constraint type in natural order: a constraint takes the lowest colour none of its particles has
    used (within its type). Returns (types, ids, class_offsets): the flat order 'type 0 colour 0, type 0 colour 1, ..., type 1
    colour 0, ...' with ids ascending inside a class, and the offsets of the classes in it."""
    types, ids, offsets = [], [], [0]
    for t, idx in enumerate((dist_ij, vol_ijkl, bend_ijkl)):
        idx = np.asarray(idx, np.int64).reshape(-1, 2 if t == 0 else 4)
        used = [0] * n_particles           # bit mask of colours per particle (python ints: any number of colours)
        colour = np.zeros(len(idx), np.int64)
        for k, verts in enumerate(idx.tolist()):
            m = 0
            for p in verts:
                m |= used[p]
            c = 0
            while (m >> c) & 1:
                c += 1
            for p in verts:
                used[p] |= 1 << c
            colour[k] = c
        for c in range(int(colour.max()) + 1 if len(idx) else 0):
            members = np.nonzero(colour == c)[0]
            types.append(np.full(len(members), t, np.uint8)); ids.append(members.astype(np.int32))
            offsets.append(offsets[-1] + len(members))
    if not ids:
        return np.zeros(0, np.uint8), np.zeros(0, np.int32), np.array([0], np.int64)
    return np.concatenate(types), np.concatenate(ids), np.array(offsets, np.int64)


class NumpySolver:
    """SPEC.md, vectorised by colour class. State: x, v as three component arrays each; w inverse masses."""

    def __init__(self, pos, vel, inv_mass, gravity=(0.0, -9.81, 0.0), damping=0.0):
        pos = np.asarray(pos, F).reshape(-1, 3)
        self.n = pos.shape[0]
        self.X = [pos[:, c].copy() for c in range(3)]
        vel = np.zeros_like(pos) if vel is None else np.asarray(vel, F).reshape(-1, 3)
        self.V = [vel[:, c].copy() for c in range(3)]
        self.P = [np.zeros(self.n, F) for _ in range(3)]         # xprev
        self.w = np.asarray(inv_mass, F).reshape(-1).copy()
        self.g = tuple(F(c) for c in gravity)
        self.d = F(damping)
        self.alpha = [F(0), F(0), F(0)]
        self.dist = self.vol = self.bend = None
        self.plane = None
        self.classes = []          # (type, ids) per colour class, in execution order
        self.classes_odd = None    # optional second list for the odd substeps of a tick (SPEC §3: one order per substep parity)

    # ---- authoring --------------------------------------------------------------------------------
    def set_distance(self, ij, rest, compliance=0.0):
        self.dist = (np.asarray(ij, np.int64).reshape(-1, 2), np.asarray(rest, F).reshape(-1)); self.alpha[0] = F(compliance)

    def set_volume(self, ijkl, rest_vol, compliance=0.0):
        # SPEC §5: the stored rest value is R6 = 6*V0, one binary32 product at set time
        self.vol = (np.asarray(ijkl, np.int64).reshape(-1, 4), F(6.0) * np.asarray(rest_vol, F).reshape(-1)); self.alpha[1] = F(compliance)

    def set_bending(self, ijkl, rest_cs, compliance=0.0):
        self.bend = (np.asarray(ijkl, np.int64).reshape(-1, 4), np.asarray(rest_cs, F).reshape(-1, 2)); self.alpha[2] = F(compliance)

    def set_ground_plane(self, normal, d):
        self.plane = (F(normal[0]), F(normal[1]), F(normal[2]), F(d))

    def set_classes(self, types, ids, offsets, parity=None):
        """Classes = slices [offsets[k], offsets[k+1]) of the flat order; a slice that mixes types is split by type (its members
        share no particle, so their relative order is free). parity=1 sets the list of the odd substeps only."""
        out = []
        types = np.asarray(types); ids = np.asarray(ids, np.int64)
        for a, b in zip(offsets[:-1], offsets[1:]):
            for t in range(3):
                members = ids[a:b][types[a:b] == t]
                if len(members):
                    out.append((t, members))
        if parity == 1:
            self.classes_odd = out
        else:
            self.classes = out

    @property
    def x(self):
        return np.stack(self.X, axis=1)

    @property
    def v(self):
        return np.stack(self.V, axis=1)

    def set_kinematic_positions(self, ids, pos):
        """SPEC.md §2, kinematic particles: between two ticks x <- target for particles with w = 0 (others are refused)."""
        ids = np.asarray(ids, np.int64); pos = np.asarray(pos, np.float32).reshape(-1, 3)
        if (self.w[ids] != 0).any():
            raise ValueError("only particles with inverse mass 0 are kinematic")
        for c in range(3):
            self.X[c][ids] = pos[:, c]

    # ---- SPEC §2 ----------------------------------------------------------------------------------
    def step(self, dt, substeps):
        S = F(substeps)
        h = F(dt) / S
        inv_h = F(1.0) / h
        hg = [h * gc for gc in self.g]
        kd = F(1.0) - self.d * h
        if kd < 0:
            kd = F(0.0)
        h2 = h * h
        at = [self.alpha[0] / h2, F(36.0) * (self.alpha[1] / h2), self.alpha[2] / h2]
        free = self.w > 0
        for sub in range(substeps):
            for c in range(3):                                   # 1. integrate
                self.P[c] = self.X[c].copy()
                vc = self.V[c] + hg[c]
                self.V[c] = np.where(free, vc, self.V[c])
                self.X[c] = np.where(free, self.X[c] + h * self.V[c], self.X[c])
            for t, ids in (self.classes_odd if (sub & 1) and self.classes_odd is not None else self.classes):     # 2. project, class by class
                if t == 0:
                    self._distance(ids, at[0])
                elif t == 1:
                    self._volume(ids, at[1])
                else:
                    self._bending(ids, at[2])
            if self.plane is not None:                           # 2b. collide
                nx, ny, nz, pd = self.plane
                pen = ((nx * self.X[0] + ny * self.X[1]) + nz * self.X[2]) - pd
                hit = free & (pen < 0)
                for c, nc in enumerate((nx, ny, nz)):
                    self.X[c] = np.where(hit, self.X[c] - pen * nc, self.X[c])
            for c in range(3):                                   # 3. velocity
                self.V[c] = ((self.X[c] - self.P[c]) * inv_h) * kd

    # ---- helpers on component triples ---------------------------------------------------------------
    @staticmethod
    def _dot(a, b):
        return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]

    @staticmethod
    def _cross(a, b):
        return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]

    def _get(self, p):
        return [self.X[c][p] for c in range(3)]

    def _put(self, p, ok, new):
        for c in range(3):
            self.X[c][p[ok]] = new[c][ok]

    # ---- SPEC §4 ----------------------------------------------------------------------------------
    def _distance(self, ids, at):
        ij, rest = self.dist
        i, j = ij[ids, 0], ij[ids, 1]
        wi, wj = self.w[i], self.w[j]
        xi, xj = self._get(i), self._get(j)
        d = [xi[c] - xj[c] for c in range(3)]
        L2 = self._dot(d, d)
        ws = (wi + wj) + at
        ok = (L2 >= _TINY) & (ws > 0)
        with np.errstate(all="ignore"):
            L = np.sqrt(L2)
            C = L - rest[ids]
            s = (-C) / (ws * L)
            si, sj = wi * s, wj * s
            self._put(i, ok, [xi[c] + si * d[c] for c in range(3)])
            self._put(j, ok, [xj[c] - sj * d[c] for c in range(3)])

    # ---- SPEC §5 ----------------------------------------------------------------------------------
    def _volume(self, ids, at):
        q, R6 = self.vol
        p = [q[ids, k] for k in range(4)]
        x = [self._get(pk) for pk in p]
        w = [self.w[pk] for pk in p]
        e1 = [x[1][c] - x[0][c] for c in range(3)]; e2 = [x[2][c] - x[0][c] for c in range(3)]; e3 = [x[3][c] - x[0][c] for c in range(3)]
        g1 = self._cross(e2, e3); g2 = self._cross(e3, e1); g3 = self._cross(e1, e2)
        g0 = [-((g1[c] + g2[c]) + g3[c]) for c in range(3)]
        g = [g0, g1, g2, g3]
        C6 = self._dot(e1, g1) - R6[ids]
        den = (((w[0] * self._dot(g0, g0) + w[1] * self._dot(g1, g1)) + w[2] * self._dot(g2, g2)) + w[3] * self._dot(g3, g3)) + at
        ok = den > 0
        with np.errstate(all="ignore"):
            s = (-C6) / den
            for k in range(4):
                ws = w[k] * s
                self._put(p[k], ok, [x[k][c] + ws * g[k][c] for c in range(3)])

    # ---- SPEC §6 ----------------------------------------------------------------------------------
    def _bending(self, ids, at):
        q, rest = self.bend
        p = [q[ids, k] for k in range(4)]
        a, b, c, d = (self._get(pk) for pk in p)
        w = [self.w[pk] for pk in p]
        sub = lambda u, v: [u[k] - v[k] for k in range(3)]
        e = sub(b, a)
        el2 = self._dot(e, e)
        with np.errstate(all="ignore"):
            el = np.sqrt(el2)
            n1 = self._cross(sub(a, c), sub(b, c)); n2 = self._cross(sub(b, d), sub(a, d))
            q1 = self._dot(n1, n1); q2 = self._dot(n2, n2)
            ok = (el > 0) & (q1 > 0) & (q2 > 0)
            m1 = [n1[k] / q1 for k in range(3)]; m2 = [n2[k] / q2 for k in range(3)]
            gc = [el * m1[k] for k in range(3)]; gd = [el * m2[k] for k in range(3)]
            ta1 = self._dot(sub(c, b), e) / el; ta2 = self._dot(sub(d, b), e) / el
            tb1 = self._dot(sub(a, c), e) / el; tb2 = self._dot(sub(a, d), e) / el
            ga = [ta1 * m1[k] + ta2 * m2[k] for k in range(3)]
            gb = [tb1 * m1[k] + tb2 * m2[k] for k in range(3)]
            s1 = np.sqrt(q1); s2 = np.sqrt(q2)
            u1 = [n1[k] / s1 for k in range(3)]; u2 = [n2[k] / s2 for k in range(3)]
            cs = self._dot(u1, u2)
            sn = -(self._dot(self._cross(u1, u2), e) / el)
            C = sn * rest[ids, 0] - cs * rest[ids, 1]
            den = (((w[0] * self._dot(ga, ga) + w[1] * self._dot(gb, gb)) + w[2] * self._dot(gc, gc)) + w[3] * self._dot(gd, gd)) + at
            ok = ok & (den > 0)
            s = (-C) / den
            for pk, xk, wk, gk in zip(p, (a, b, c, d), w, (ga, gb, gc, gd)):
                ws = wk * s
                self._put(pk, ok, [xk[k] + ws * gk[k] for k in range(3)])
