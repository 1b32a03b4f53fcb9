"""Second, independent restatement of DIRECT's IPDDP optimiser in NumPy -- TEST INFRASTRUCTURE.

Where oracle/direct_ref.c is written with scalar loops, this file follows the Eigen expressions of
global_planner/src/ddp_optimizer.cpp ("DDP") literally: the same dense helper matrices
(hatAbarpoly2minvotau, barpoly2minvotau_v, temp, tempv, ...), built with the same block fills, and
the same matrix products.  It exists only to cross-validate the C oracle and to generate the golden
fixtures under tests/golden/ (tests/golden/make_golden.py).  PARITY UNPINNED: see direct_ref.c.
"""
import numpy as np

MINVO6 = np.array([
    [1.0, -0.06471861202, -0.03728008486, -0.02577637794, -0.02027573243, -0.01678273037],
    [1.0, 0.03314986096, -0.06548114211, -0.05530463802, -0.04362718953, -0.03671639115],
    [1.0, 0.3375528997, 0.05836232552, -0.02920033165, -0.04690387913, -0.04376447947],
    [1.0, 0.6624471003, 0.3832565261, 0.1916286091, 0.06985980172, -0.002892843108],
    [1.0, 0.966850139, 0.868219136, 0.7594116288, 0.6521050661, 0.5510660979],
    [1.0, 1.064718612, 1.092157139, 1.108091959, 1.118023718, 1.123960059]])  # DDP:63-68
MINVO_V6 = np.array([
    [0, 1.0, -0.1423379297, -0.1332742327, -0.1242105357, -0.126304257],
    [0, 1.0, 0.1887439858, -0.1831318297, -0.2466606848, -0.2321393311],
    [0, 1.0, 1.0, 0.5411016575, 0.08220331498, -0.2433474658],
    [0, 1.0, 1.811256014, 2.250636213, 2.381669451, 2.282405938],
    [0, 1.0, 2.14233793, 3.293739556, 4.445141183, 5.585385392]])  # DDP:69-73
MINVO_A6 = np.array([
    [0, 0, 2.0, -0.4472869252, -0.6133793313, -0.6406553622],
    [0, 0, 2.0, 1.223711659, -0.5552714346, -1.854618819],
    [0, 0, 2.0, 4.776288341, 6.54988193, 6.841145057],
    [0, 0, 2.0, 6.447286925, 13.17576837, 22.04662796]])  # DDP:74-77
BEZ6 = np.array([[1.0, 0, 0, 0, 0, 0], [1.0, 0.2, 0, 0, 0, 0], [1.0, 0.4, 0.1, 0, 0, 0],
                 [1.0, 0.6, 0.3, 0.1, 0, 0], [1.0, 0.8, 0.6, 0.4, 0.2, 0],
                 [1.0, 1.0, 1.0, 1.0, 1.0, 1.0]])  # DDP:79-84
BEZ_V6 = np.array([[0, 1.0, 0, 0, 0, 0], [0, 1.0, 0.5, 0, 0, 0], [0, 1.0, 1.0, 0.5, 0, 0],
                   [0, 1.0, 1.5, 1.5, 1.0, 0], [0, 1.0, 2.0, 3.0, 4.0, 5.0]])  # DDP:86-90
BEZ_A6 = np.array([[0, 0, 2.0, 0, 0, 0], [0, 0, 2.0, 2.0, 0, 0], [0, 0, 2.0, 4.0, 4.0, 0],
                   [0, 0, 2.0, 6.0, 12.0, 20.0]])  # DDP:92-95


def _dt_tables(Tk):
    """poly2minvotau_dt, _v_dt, _a_dt of DDP:1543-1561 (num_ctrlP == 6)."""
    Tk2, Tk3, Tk4 = Tk * Tk, Tk ** 3, Tk ** 4
    d6 = np.array([
        [0, -0.06471861202, -0.07456016972 * Tk, -0.07732913382 * Tk2, -0.08110292972 * Tk3, -0.08391365186 * Tk4],
        [0, 0.03314986096, -0.1309622842 * Tk, -0.1659139141 * Tk2, -0.1745087581 * Tk3, -0.1835819558 * Tk4],
        [0, 0.3375528997, 0.116724651 * Tk, -0.08760099494 * Tk2, -0.1876155165 * Tk3, -0.2188223973 * Tk4],
        [0, 0.6624471003, 0.7665130522 * Tk, 0.5748858272 * Tk2, 0.2794392069 * Tk3, -0.01446421554 * Tk4],
        [0, 0.966850139, 1.736438272 * Tk, 2.278234886 * Tk2, 2.608420264 * Tk3, 2.755330489 * Tk4],
        [0, 1.064718612, 2.184314278 * Tk, 3.324275878 * Tk2, 4.472094873 * Tk3, 5.619800295 * Tk4]])
    dv = np.array([
        [0, 0, -0.1423379297, -0.2665484655 * Tk, -0.3726316072 * Tk2, -0.5052170278 * Tk3],
        [0, 0, 0.1887439858, -0.3662636595 * Tk, -0.7399820545 * Tk2, -0.9285573245 * Tk3],
        [0, 0, 1.0, 1.082203315 * Tk, 0.2466099449 * Tk2, -0.9733898632 * Tk3],
        [0, 0, 1.811256014, 4.501272426 * Tk, 7.145008354 * Tk2, 9.129623752 * Tk3],
        [0, 0, 2.14233793, 6.587479113 * Tk, 13.33542355 * Tk2, 22.34154157 * Tk3]])
    da = np.array([
        [0, 0, 0, -0.4472869252, -1.226758663 * Tk, -1.921966087 * Tk2],
        [0, 0, 0, 1.223711659, -1.110542869 * Tk, -5.563856457 * Tk2],
        [0, 0, 0, 4.776288341, 13.09976386 * Tk, 20.52343517 * Tk2],
        [0, 0, 0, 6.447286925, 26.35153674 * Tk, 66.13988387 * Tk2]])
    return d6, dv, da


TEMPM = np.array([[1, 0, 0, 0, 0, 0], [-5, 5, 0, 0, 0, 0], [10, -20, 10, 0, 0, 0],
                  [-10, 30, -30, 10, 0, 0], [5, -20, 30, -20, 5, 0], [-1, 5, -10, 10, -5, 1]], float)  # DDP:1050-1055
I3 = np.eye(3)


def llt(A):
    """Eigen::LLT semantics: lower triangle only; fail iff a pivot <= 0 (NaN passes)."""
    n = A.shape[0]
    L = np.tril(A).astype(float)
    for k in range(n):
        x = L[k, k] - L[k, :k] @ L[k, :k]
        if x <= 0.0:
            return None
        x = np.sqrt(x)
        L[k, k] = x
        if k + 1 < n:
            L[k + 1:, k] = (L[k + 1:, k] - L[k + 1:, :k] @ L[k, :k]) / x
    return L


def llt_solve(L, B):
    n = L.shape[0]
    Y = np.array(B, float)
    for i in range(n):
        Y[i] = (Y[i] - L[i, :i] @ Y[:i]) / L[i, i]
    for i in range(n - 1, -1, -1):
        Y[i] = (Y[i] - L[i + 1:, i] @ Y[i + 1:]) / L[i, i]
    return Y


class DDP:
    """ddpTrajOptimizer + fwdPass + bwdPass + algParam (ddp_optimizer.h:18-342) for one corridor."""

    def __init__(self, planes, durations, pos, vel, acc, max_vel, max_acc, initbez, w_snap,
                 w_terminal, w_time, iter_max, infeas, zero_init, line_init=False, time_power=2,
                 minvo=False, seeds=None, fixed_iters=False):
        # DDP:33-61
        self.planes = [np.asarray(p, float).reshape(-1, 4) for p in planes]
        self.N = N = len(self.planes)
        self.maxiter, self.tol, self.infeas = iter_max, 1.0e-7, bool(infeas)
        self.infeas_ref, self.line_failed = bool(infeas), True
        self.w_snap, self.Rtime, self.time_power = w_snap, w_time, time_power
        self.maxVel, self.maxAcc, self.minvo = max_vel, max_acc, bool(minvo)
        self.zero_init, self.line_init, self.fixed_iters = bool(zero_init), bool(line_init), fixed_iters
        self.reg_exp_base = 1.6 if zero_init else 4.0
        self.M6, self.Mv6, self.Ma6 = (MINVO6, MINVO_V6, MINVO_A6) if minvo else (BEZ6, BEZ_V6, BEZ_A6)
        self.Ek_inv = [1.0, 1.0, 0.5]
        self.barEk_inv = np.repeat(self.Ek_inv, 3)  # DDP:181-186
        pos, vel, acc = (np.asarray(a, float).reshape(2, 3) for a in (pos, vel, acc))
        self.x_d = np.concatenate([pos[1], vel[1], acc[1]])  # DDP:104-111
        self.Pmat = w_terminal * np.eye(9)
        x0 = np.concatenate([pos[0], vel[0], acc[0]])
        self.x = [x0] + [np.zeros(9) for _ in range(N)]
        self.u, self.s, self.y, self.c = [], [], [], []
        for i in range(N):  # DDP:124-160
            ui = np.zeros(10)
            ui[9] = durations[i]
            self.u.append(ui)
            nc = self.planes[i].shape[0] * 6 + 5 * 3 * 2 + 4 * 3 * 2 + 1
            self.c.append(np.zeros(nc))
            self.s.append(0.1 * np.ones(nc))
            self.y.append(0.01 * np.ones(nc))
        # DDP:154-159: the gains start as zeros (what a forwardpass() after a stuck first backwardpass() steps with)
        self.ku = [np.zeros(10) for _ in range(N)]; self.Ku = [np.zeros((10, 9)) for _ in range(N)]
        self.ks = [np.zeros(ci.size) for ci in self.c]; self.Ks = [np.zeros((ci.size, 9)) for ci in self.c]
        self.ky = [np.zeros(ci.size) for ci in self.c]; self.Ky = [np.zeros((ci.size, 9)) for ci in self.c]
        self.PolyTime = np.array(durations, float)
        if not zero_init:
            if not line_init:
                # DDP:167-193
                BezCoeff = np.array(initbez, float).reshape(N, 18)
                il = np.stack([BezCoeff[i].reshape(3, 6).T.reshape(-1) for i in range(N)])  # Map(6,3).T flattened
                poly = self.bez2poly(il, self.PolyTime)
                for i in range(N):
                    self.u[i][:9] = poly[i, 9:]
            else:
                self._line_init(pos, seeds)
        self.initialroll()
        if line_init:  # DDP:255-269
            if sum(int((ci > 0).sum()) for ci in self.c) == 0:
                self.infeas = False
        self.costTraj, self.costqTraj = [self.cost], [self.costq]
        self.mu = self.cost / self.N / self.s[0].size  # DDP:281
        self.resetfilter()
        self.reg, self.bp_failed = 0.0, False
        if line_init:
            self.reg = 10.0
        self.rtn, self.iter, self.fwd_passes = 0, 0, 0
        self.bp_no_upd_count, self.no_upd_count = 0, 0
        self.opterr = 0.0
        self.trace = []

    # ---- tables --------------------------------------------------------------------------
    @staticmethod
    def FG(Tk):  # DDP:836-890
        F = np.array([[1.0, Tk, Tk * Tk / 2.0], [0, 1.0, Tk], [0, 0, 1.0]])
        G = np.array([[Tk ** 3, Tk ** 4, Tk ** 5], [3 * Tk ** 2, 4 * Tk ** 3, 5 * Tk ** 4], [6 * Tk, 12 * Tk ** 2, 20 * Tk ** 3]])
        return np.kron(F, I3), np.kron(G, I3)

    @staticmethod
    def FGprime(Tk):  # DDP:892-962
        Fp = np.array([[0, 1.0, Tk], [0, 0, 1.0], [0, 0, 0]])
        Gp = np.array([[3 * Tk ** 2, 4 * Tk ** 3, 5 * Tk ** 4], [6 * Tk, 12 * Tk ** 2, 20 * Tk ** 3], [6, 24 * Tk, 60 * Tk ** 2]])
        return np.kron(np.triu(Fp, 1), I3), np.kron(Gp, I3)

    @staticmethod
    def Rmats(Tk):  # DDP:964-1015
        R = np.array([[36 * Tk, 72 * Tk ** 2, 120 * Tk ** 3], [72 * Tk ** 2, 192 * Tk ** 3, 360 * Tk ** 4], [120 * Tk ** 3, 360 * Tk ** 4, 720 * Tk ** 5]])
        Rp = np.array([[36, 144 * Tk, 360 * Tk ** 2], [144 * Tk, 576 * Tk ** 2, 1440 * Tk ** 3], [360 * Tk ** 2, 1440 * Tk ** 3, 3600 * Tk ** 4]])
        Rpp = np.array([[0, 144, 720 * Tk], [144, 1152 * Tk, 4320 * Tk ** 2], [720 * Tk, 4320 * Tk ** 2, 14400 * Tk ** 3]])
        return np.kron(R, I3), np.kron(Rp, I3), np.kron(Rpp, I3)

    @staticmethod
    def beztau2polyt(T):  # DDP:1018-1059, 788
        return TEMPM.T @ np.diag([(1.0 / T) ** i for i in range(6)])

    def bez2poly(self, il, T):  # DDP:782-796
        out = np.zeros_like(il)
        for i in range(il.shape[0]):
            Bm = (T[i] * il[i]).reshape(6, 3).T  # Map(3,6) column-major
            out[i] = (Bm @ self.beztau2polyt(T[i])).T.reshape(-1)
        return out

    def poly2bez(self, poly, T):  # DDP:799-812
        out = np.zeros_like(poly)
        for i in range(poly.shape[0]):
            Pm = (1.0 / T[i] * poly[i]).reshape(6, 3).T
            out[i] = (Pm @ np.linalg.inv(self.beztau2polyt(T[i]))).T.reshape(-1)
        return out

    # ---- model ---------------------------------------------------------------------------
    def computenextx(self, x, u):  # DDP:1062-1067
        bF, bG = self.FG(u[9])
        return np.triu(bF) @ x + bG @ u[:9]

    def _scaled(self, T):
        Tkv = np.array([T ** (i + 1) for i in range(7)])
        p6 = np.zeros((6, 6)); p6[:, 0] = 1.0
        for i in range(1, 6):
            p6[:, i] = self.M6[:, i] * Tkv[i - 1]
        v6 = np.zeros((5, 6)); v6[:, 1] = self.Mv6[:, 1]
        for i in range(2, 6):
            v6[:, i] = self.Mv6[:, i] * Tkv[i - 2]
        a6 = np.zeros((4, 6)); a6[:, 2] = self.Ma6[:, 2]
        for i in range(3, 6):
            a6[:, i] = self.Ma6[:, i] * Tkv[i - 3]
        return p6, v6, a6

    def computecminvo(self, x, u, k, want_mats=False):  # DDP:1132-1285
        polyCoeff = np.zeros((6, 3))
        for j in range(3):
            for q in range(3):
                polyCoeff[q, j] = x[q * 3 + j] * self.Ek_inv[q]
            for q in range(3, 6):
                polyCoeff[q, j] = u[(q - 3) * 3 + j]
        p6, v6, a6 = self._scaled(u[9])
        posCoeff = p6 @ polyCoeff
        pl = self.planes[k]
        P = pl.shape[0]
        c_temp = np.zeros(P * 6)
        for j in range(6):
            for q in range(P):
                c_temp[j * P + q] = pl[q, 0] * posCoeff[j, 0] + pl[q, 1] * posCoeff[j, 1] + pl[q, 2] * posCoeff[j, 2] + pl[q, 3]
        barEkinv = np.diag(self.barEk_inv)
        tempv = np.concatenate([barEkinv @ x, u[:9]])
        bar_v = np.kron(v6, I3)  # (j*3+l, k*3+l) = v6(j,k); column 0 of v6 is zero (DDP:1228-1234)
        bar_a = np.kron(a6, I3)
        tempcv = bar_v @ tempv
        c_v = np.concatenate([tempcv - self.maxVel, -tempcv - self.maxVel])
        tempca = bar_a @ tempv
        c_a = np.concatenate([tempca - self.maxAcc, -tempca - self.maxAcc])
        c = np.concatenate([c_temp, c_v, c_a, [-u[9] + 0.3]])
        if not self.minvo:
            c = c - 2.0e-4
        if want_mats:
            return c, p6, bar_v, bar_a, tempv
        return c

    def computep(self, x):  # DDP:1289-1292
        return 0.5 * float((x - self.x_d) @ self.Pmat @ (x - self.x_d))

    def computeq(self, u):  # DDP:1294-1305
        bR, _, _ = self.Rmats(u[9])
        if self.time_power == 2:
            return 0.5 * self.w_snap * float(u[:9] @ bR @ u[:9]) + 0.5 * u[9] * self.Rtime * u[9]
        return 0.5 * self.w_snap * float(u[:9] @ bR @ u[:9]) + 0.5 * self.Rtime * u[9]

    def computeall(self):  # DDP:1309-1604
        N = self.N
        self.p = self.computep(self.x[N])
        self.px = self.Pmat @ (self.x[N] - self.x_d)
        self.pxx = self.Pmat.copy()
        self.fx, self.fu, self.qu, self.quu, self.cx, self.cu = [], [], [], [], [], []
        for i in range(N):
            x, u = self.x[i], self.u[i]
            bF, bG = self.FG(u[9])
            bFp, bGp = self.FGprime(u[9])
            fgradt = bFp @ x + bGp @ u[:9]
            self.fx.append(bF)
            self.fu.append(np.hstack([bG, fgradt[:, None]]))
            bR, bRp, bRpp = self.Rmats(u[9])
            ub = u[:9]
            if self.time_power == 2:
                qu = np.concatenate([self.w_snap * bR @ ub, [self.Rtime * u[9] + 0.5 * self.w_snap * ub @ bRp @ ub]])
                corner = self.Rtime + 0.5 * self.w_snap * float(ub @ bRpp @ ub)
            else:
                qu = np.concatenate([self.w_snap * bR @ ub, [0.5 * self.Rtime + 0.5 * self.w_snap * ub @ bRp @ ub]])
                corner = 0.5 * self.w_snap * float(ub @ bRpp @ ub)
            quu = np.block([[self.w_snap * bR, (self.w_snap * bRp @ ub)[:, None]],
                            [(self.w_snap * ub @ bRp)[None, :], np.array([[corner]])]])
            self.qu.append(qu)
            self.quu.append(quu)
            # DDP:1455-1604
            c, p6, bar_v, bar_a, tempv = self.computecminvo(x, u, i, want_mats=True)
            self.c[i] = c
            pl = self.planes[i]
            P = pl.shape[0]
            hatA = np.zeros((6 * P, 18))
            for j in range(6):
                for k in range(6):
                    for ld in range(P):
                        hatA[j * P + ld, k * 3:k * 3 + 3] = p6[j, k] * pl[ld, :3]
            temp = np.zeros((18, 18))
            temp[:9, :9] = np.diag(self.barEk_inv)
            temp[9:, 9:] = np.eye(9)
            tempcxv = bar_v @ temp[:, :9]
            tempcxa = bar_a @ temp[:, :9]
            cx = np.vstack([hatA @ temp[:, :9], tempcxv, -tempcxv, tempcxa, -tempcxa, np.zeros((1, 9))])
            d6, dv, da = _dt_tables(u[9])
            hatA_dt = np.zeros((6 * P, 18))
            for j in range(6):
                for k in range(6):
                    for ld in range(P):
                        hatA_dt[j * P + ld, k * 3:k * 3 + 3] = d6[j, k] * pl[ld, :3]
            bar_v_dt = np.kron(dv, I3)
            bar_a_dt = np.kron(da, I3)
            tempcuv = bar_v @ temp[:, 9:]
            tempcuv_time = bar_v_dt @ tempv
            tempcua = bar_a @ temp[:, 9:]
            tempcua_time = bar_a_dt @ tempv
            cu = np.vstack([
                np.hstack([hatA @ temp[:, 9:], (hatA_dt @ tempv)[:, None]]),
                np.hstack([tempcuv, tempcuv_time[:, None]]),
                np.hstack([-tempcuv, -tempcuv_time[:, None]]),
                np.hstack([tempcua, tempcua_time[:, None]]),
                np.hstack([-tempcua, -tempcua_time[:, None]]),
                np.hstack([np.zeros((1, 9)), [[-1.0]]])])
            self.cx.append(cx)
            self.cu.append(cu)

    def initialroll(self):  # DDP:1608-1620
        self.q = np.zeros(self.N)
        for i in range(self.N):
            self.c[i] = self.computecminvo(self.x[i], self.u[i], i)
            self.q[i] = self.computeq(self.u[i])
            self.x[i + 1] = self.computenextx(self.x[i], self.u[i])
        self.cost = self.q.sum() + self.computep(self.x[self.N])
        self.costq = self.q.sum()

    def resetfilter(self):  # DDP:1636-1662
        logcost, err = self.cost, 0.0
        with np.errstate(all="ignore"):
            if self.infeas:
                for i in range(self.N):
                    logcost -= self.mu * np.log(self.y[i]).sum()
                    err += np.abs(self.c[i] + self.y[i]).sum()
                if err < self.tol:
                    err = 0.0
            else:
                for i in range(self.N):
                    logcost -= self.mu * np.log(-self.c[i]).sum()
        self.logcost, self.err = logcost, err
        self.filter = [(logcost, err)]
        self.step, self.fp_failed = 0, False

    def _line_init(self, pos, seeds):  # DDP:194-248
        N = self.N
        points = [pos[0]] + [np.asarray(seeds[i], float) for i in range(1, N)] + [pos[1]]
        for l in range(N):
            vio, cnt = True, 0
            while vio and cnt <= 4:
                Tk = self.u[l][9]
                Fk = np.array([[1.0, Tk, Tk * Tk / 2.0], [0, 1.0, Tk], [0, 0, 1.0]])
                Gi = np.array([[10.0 / Tk ** 3, -4.0 / Tk ** 2, 0.5 / Tk], [-15.0 / Tk ** 4, 7.0 / Tk ** 3, -1.0 / Tk ** 2],
                               [6.0 / Tk ** 5, -3.0 / Tk ** 4, 0.5 / Tk ** 3]])
                xn, xc = np.zeros(9), np.zeros(9)
                xn[:3], xc[:3] = points[l + 1], points[l]
                self.u[l][:9] = np.kron(Gi, I3) @ (xn - np.kron(Fk, I3) @ xc)
                cons = self.computecminvo(xc, self.u[l], l)
                if (cons < 0).all():
                    vio = False
                else:
                    self.u[l][9] = 2 * Tk
                    cnt += 1

    # ---- passes --------------------------------------------------------------------------
    def backwardpass(self):  # DDP:440-644
        N = self.N
        if self.fp_failed or self.bp_failed:
            self.reg += 1.0
        elif self.step == 0:
            self.reg -= 1.0
        elif self.step <= 3:
            pass
        else:
            self.reg += 1.0
        self.reg = min(max(self.reg, 0.0), 24.0)
        if not self.fp_failed:
            self.computeall()
        Vx, Vxx = self.px.copy(), self.pxx.copy()
        c_err = mu_err = Qu_err = 0.0
        for i in range(N - 1, -1, -1):
            fx, fu, cx, cu, c, s, y = self.fx[i], self.fu[i], self.cx[i], self.cu[i], self.c[i], self.s[i], self.y[i]
            Qx = cx.T @ s + fx.T @ Vx
            Qu = self.qu[i] + cu.T @ s + fu.T @ Vx
            fxiVxx = fx.T @ Vxx
            Qxx = fxiVxx @ fx
            Qxu = fxiVxx @ fu
            Quu = self.quu[i] + fu.T @ Vxx @ fu
            Quu = 0.5 * (Quu + Quu.T)
            Quu_reg = Quu + (self.reg_exp_base ** self.reg - 1) * np.eye(10)
            if self.infeas:
                r = s * y - self.mu
                rhat = s * (c + y) - r
                yinv = 1.0 / y
                SYinv = np.diag(s * yinv)
                cuitSYinvcui = cu.T @ SYinv @ cu
                SYinvcxi = SYinv @ cx
                L = llt(Quu_reg + cuitSYinvcui)
                if L is None:
                    self.bp_failed, self.opterr = True, np.inf
                    return
                tempv2 = yinv * rhat
                Qu = Qu + cu.T @ tempv2
                tempQux = Qxu.T + cu.T @ SYinvcxi
                kK = -llt_solve(L, np.hstack([Qu[:, None], tempQux]))
                ku, Ku = kK[:, 0], kK[:, 1:]
                cuiku = cu @ ku
                cxiPluscuiKu = cx + cu @ Ku
                self.ks[i] = yinv * (rhat + s * cuiku)
                self.Ks[i] = SYinv @ cxiPluscuiKu
                self.ky[i] = -(c + y) - cuiku
                self.Ky[i] = -cxiPluscuiKu
                Quu = Quu + cuitSYinvcui
                Qxu = tempQux.T
                Qxx = Qxx + cx.T @ SYinvcxi
                Qx = Qx + cx.T @ tempv2
            else:
                r = s * c + self.mu
                cinv = 1.0 / c
                SCinv = np.diag(s * cinv)
                SCinvcui = SCinv @ cu
                SCinvcxi = SCinv @ cx
                cuitSCinvcui = cu.T @ SCinvcui
                L = llt(Quu_reg - cuitSCinvcui)
                if L is None:
                    self.bp_failed, self.opterr = True, np.inf
                    return
                tempv2 = cinv * r
                Qu = Qu - cu.T @ tempv2
                tempQux = Qxu.T - cu.T @ SCinvcxi
                kK = -llt_solve(L, np.hstack([Qu[:, None], tempQux]))
                ku, Ku = kK[:, 0], kK[:, 1:]
                cuiku = cu @ ku
                self.ks[i] = -(cinv * (r + s * cuiku))
                self.Ks[i] = -(SCinv @ (cx + cu @ Ku))
                self.ky[i] = np.zeros(c.size)
                self.Ky[i] = np.zeros((c.size, 9))
                Quu = Quu - cuitSCinvcui
                Qxu = tempQux.T
                Qxx = Qxx - cx.T @ SCinvcxi
                Qx = Qx - cx.T @ tempv2
            QxuKu = Qxu @ Ku
            KutQuu = Ku.T @ Quu
            Vx = Qx + Ku.T @ Qu + KutQuu @ ku + Qxu @ ku
            Vxx = Qxx + QxuKu.T + QxuKu + KutQuu @ Ku
            Vxx = 0.5 * (Vxx + Vxx.T)
            self.ku[i], self.Ku[i] = ku, Ku
            Qu_err = max(Qu_err, np.abs(Qu).max())
            mu_err = max(mu_err, np.abs(r).max())
            if self.infeas:
                c_err = max(c_err, np.abs(c + y).max())
        self.bp_failed = False
        self.opterr = max(Qu_err, c_err, mu_err)

    def forwardpass(self):  # DDP:647-778
        N = self.N
        xold, uold, yold, sold, cold = self.x, self.u, self.y, self.s, self.c
        xnew = [a.copy() for a in xold]; unew = [a.copy() for a in uold]
        cnew = [a.copy() for a in cold]; ynew = [a.copy() for a in yold]; snew = [a.copy() for a in sold]
        tau = max(0.99, 1 - self.mu)
        failed = False
        for step in range(11):
            failed = False
            stepsize = 2.0 ** (-step)
            xnew[0] = xold[0].copy()
            for i in range(N):
                dx = xnew[i] - xold[i]
                if self.infeas:
                    ynew[i] = yold[i] + stepsize * self.ky[i] + self.Ky[i] @ dx
                    snew[i] = sold[i] + stepsize * self.ks[i] + self.Ks[i] @ dx
                    if (ynew[i] < (1 - tau) * yold[i]).any() or (snew[i] < (1 - tau) * sold[i]).any():
                        failed = True
                        break
                    unew[i] = uold[i] + stepsize * self.ku[i] + self.Ku[i] @ dx
                else:
                    snew[i] = sold[i] + stepsize * self.ks[i] + self.Ks[i] @ dx
                    unew[i] = uold[i] + stepsize * self.ku[i] + self.Ku[i] @ dx
                    cnew[i] = self.computecminvo(xnew[i], unew[i], i)
                    if (cnew[i] > (1 - tau) * cold[i]).any() or (snew[i] < (1 - tau) * sold[i]).any():
                        failed = True
                        break
                xnew[i + 1] = self.computenextx(xnew[i], unew[i])
            if failed:
                continue
            qnew = np.array([self.computeq(unew[i]) for i in range(N)])
            cost = qnew.sum() + self.computep(xnew[N])
            costq = qnew.sum()
            logcost, err = cost, 0.0
            with np.errstate(all="ignore"):
                if self.infeas:
                    for i in range(N):
                        logcost -= self.mu * np.log(ynew[i]).sum()
                        cnew[i] = self.computecminvo(xnew[i], unew[i], i)
                        err += np.abs(cnew[i] + ynew[i]).sum()
                    err = max(self.tol, err)
                else:
                    for i in range(N):
                        cnew[i] = self.computecminvo(xnew[i], unew[i], i)
                        logcost -= self.mu * np.log(-cnew[i]).sum()
            keep = []
            for (f0, f1) in self.filter:
                if logcost >= f0 and err >= f1:
                    failed = True
                    break
                elif logcost > f0 or err > f1:
                    keep.append((f0, f1))
            if failed:
                continue
            self.filter = keep + [(logcost, err)]
            break
        if failed:
            self.fp_failed, self.stepsize = True, 0.0
        else:
            self.cost, self.costq, self.logcost = cost, costq, logcost
            self.x, self.u, self.y, self.s, self.c, self.q = xnew, unew, ynew, snew, cnew, qnew
            self.err, self.stepsize, self.step, self.fp_failed = err, stepsize, step, False

    # ---- outer loop ----------------------------------------------------------------------
    def iterate_once(self):  # DDP:295-412 body; True when the loop breaks
        n_sweeps = 0
        while True:
            self.backwardpass()
            n_sweeps += 1
            if not self.bp_failed:
                break
            if self.reg == 24 and self.bp_failed:
                self.bp_no_upd_count += 1
            else:
                self.bp_no_upd_count = 0
            if self.bp_no_upd_count > 20:
                break
        self.forwardpass()
        self.fwd_passes += 1
        self.trace.append((self.cost, self.costq, self.logcost, self.err, self.mu, self.reg, self.step,
                           self.opterr, self.stepsize, float(self.fp_failed), float(n_sweeps), float(self.bp_failed)))
        if any(u[9] < 0 for u in self.u):
            self.rtn = -3
            return True
        self.costTraj.append(self.cost)
        self.costqTraj.append(self.costq)
        if not self.fixed_iters and max(self.opterr, self.mu) <= self.tol:
            return True
        if self.opterr <= 0.2 * self.mu:
            self.mu = max(self.tol / 10.0, min(0.2 * self.mu, self.mu ** 1.2))
            self.resetfilter()
            self.reg, self.bp_failed = 0.0, False
        count = sum(int((ci >= 2.0e-4).sum()) for ci in self.c)
        if count == 0 and not self.fixed_iters:
            if self.zero_init:
                self.infeas_ref, self.rtn = False, 2
                return True
            if not self.zero_init and not self.line_init:
                if (self.cost - self.costTraj[-2]) ** 2 < self.costTraj[-2] * 1.0e-2 and self.opterr < 5.0e1:
                    self.rtn = 1
                    return True
            if self.line_init:
                if (self.cost - self.costTraj[-2]) ** 2 < self.costTraj[-2] * 0.01:
                    self.line_failed = False
                    return True
        if self.bp_no_upd_count > 20:
            self.rtn = -4
            return True
        if self.line_init:
            self.no_upd_count = self.no_upd_count + 1 if self.stepsize < 1.0e-6 else 0
            if self.no_upd_count > 100:
                return True
        return False

    def run(self):
        self.iter = 0
        while self.iter < self.maxiter:
            if self.iterate_once():
                break
            self.iter += 1
        return self.rtn

    def results(self):  # DDP:414-437 + getters
        N = self.N
        jerk = 0.0
        for i in range(N):
            bR, _, _ = self.Rmats(self.u[i][9])
            jerk += float(self.u[i][:9] @ bR @ self.u[i][:9])
        poly = np.zeros((N, 18))
        T = np.zeros(N)
        for i in range(N):  # sysparam2polyFunc DDP:814-823
            T[i] = self.u[i][9]
            poly[i, :9] = self.barEk_inv * self.x[i]
            poly[i, 9:] = self.u[i][:9]
        il = self.poly2bez(poly, T)
        bez = np.stack([il[i].reshape(6, 3).T.reshape(-1) for i in range(N)])  # DDP:430-436
        d = self.x[N] - self.x_d
        return dict(rtn=self.rtn, iter_used=self.iter, fwd_passes=self.fwd_passes, cost=self.cost,
                    costq=self.costq, jerk_cost=jerk, terminal_norm2=float(d @ d), opterr=self.opterr,
                    mu=self.mu, bez=bez, poly=poly, T=T, infeas_out=self.infeas_ref,
                    line_failed_out=self.line_failed)


def make_problem(batch, b, params):
    """The solver object of one problem of an abi.HostBatch, set up (DDP:5-286) but not run: for stepping."""
    N = int(batch.n_seg[b])
    planes = [batch.planes[b, k, :batch.n_planes[b, k]] for k in range(N)]
    pos = np.stack([batch.x0[b, :3], batch.xd[b, :3]])
    vel = np.stack([batch.x0[b, 3:6], batch.xd[b, 3:6]])
    acc = np.stack([batch.x0[b, 6:9], batch.xd[b, 6:9]])
    infeas = batch.infeas_in[b] if batch.infeas_in is not None else params.infeas
    return DDP(planes, batch.T0[b, :N], pos, vel, acc, params.max_vel, params.max_acc,
               None if batch.init_bez is None else batch.init_bez[b, :N], params.w_snap, params.w_terminal,
               params.w_time, params.iter_max, infeas, params.zero_init, params.line_init, params.time_power,
               params.minvo, None if batch.seeds is None else batch.seeds[b], bool(params.fixed_iters))


def solve_problem(batch, b, params):
    """Run one problem of an abi.HostBatch with an abi.Params; returns (DDP, results dict)."""
    N = int(batch.n_seg[b])
    planes = [batch.planes[b, k, :batch.n_planes[b, k]] for k in range(N)]
    pos = np.stack([batch.x0[b, :3], batch.xd[b, :3]])
    vel = np.stack([batch.x0[b, 3:6], batch.xd[b, 3:6]])
    acc = np.stack([batch.x0[b, 6:9], batch.xd[b, 6:9]])
    infeas = batch.infeas_in[b] if batch.infeas_in is not None else params.infeas
    d = DDP(planes, batch.T0[b, :N], pos, vel, acc, params.max_vel, params.max_acc,
            None if batch.init_bez is None else batch.init_bez[b, :N], params.w_snap, params.w_terminal,
            params.w_time, params.iter_max, infeas, params.zero_init, params.line_init, params.time_power,
            params.minvo, None if batch.seeds is None else batch.seeds[b], bool(params.fixed_iters))
    d.run()
    return d, d.results()
