"""CPU probe (numpy / scipy + the oracle): would an EXACT Woodbury treatment of the wells in stage 2 of the CPR preconditioner pay?
On a small 5-spot deck it compares, inside left-preconditioned GMRES(40) on the explicit-clique Jacobian, three second stages -- block-ILU0 of A
(the reservoir matrix + the wells' own-cell diagonal blocks: what the device factorises), the exact (M + L)^-1 with L = the rest of the well terms
(dense, by inversion: what a Woodbury correction would apply), and block-ILU0 of the matrix with explicit cliques -- alone and behind an exact
pressure solve (an idealised first stage).  Result (profiles/r03_probes.md): the exact correction reproduces the explicit-clique counts, and
both differ from the present preconditioner by 0-2 iterations here, mostly 0: the iterations the wells cost are not stage 2's.
    python tools/woodbury_probe.py [rate m3/day] [Newton iteration whose matrix is used]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'opm-simulators-legacy_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from opmgpu import capi, decks, wells as W
from oracle import oracle as orc
from util import OracleBackend
orc.lib()
nx,ny,nz = 14,14,10
grid = decks.cartesian_grid(nx,ny,nz, lognormal_sigma=0.5, seed=12345)
tab = decks.satfunc_standard_tables()
st = decks.initial_state(grid, tab, perturb=0.002, seed=12345)
wl = W.five_spot(grid, rate_m3_per_day=float(sys.argv[1]) if len(sys.argv)>1 else 30.0, bhp_prod_bar=150.0)
prm = capi.default_params(linear_solver_reduction=1e-10, linear_solver_maxiter=2000)
ob = OracleBackend(orc, grid, tab, prm, wells=wl.arrays())
mh = W.WellCoupledModel(ob, W.StandardWellsHost(wl, grid.z, tab.surface_density[0]), W.WellState(wl, st.p))
dt = 5*decks.DAY
mh.prepareStep(dt, st)
nc = grid.nc
def to_csr(rowptr,col,val): return sp.bsr_matrix((np.asarray(val).reshape(-1,3,3), col, rowptr), shape=(3*nc,3*nc)).tocsr()
# red-black position
ijk = np.arange(nc); i = ijk % nx; j = (ijk//nx)%ny; k = ijk//(nx*ny)
colour = (i+j+k) % 2
order = np.lexsort((ijk, colour)); position = np.empty(nc, np.int32); position[order] = np.arange(nc, dtype=np.int32)
def run_case(it_target):
    for it in range(it_target+1):
        mh.assemble(it==0) if hasattr(mh,'assemble') else None
        conv = ob.getConvergence()
        if it < it_target:
            ob.solveJacobianSystem(); mh.wh.recover_and_update(ob.perfDx(wl.nperf), mh.ws); ob.updateState()
    return
run_case(int(sys.argv[2]) if len(sys.argv)>2 else 1)
Jc = to_csr(ob.rowptr, ob.col, ob.val)                       # explicit cliques (scaled rows)
b = np.ascontiguousarray(((ob.r + ob.rhs_extra) * np.repeat(ob.scale, nc)).reshape(3,nc).T).ravel()
# reservoir-only matrix on the clique pattern
r0, val_res, _, _ = orc.assemble(grid, tab, dt, ob.st, ob.rowptr, ob.col, scale=tuple(ob.scale), accum0=ob.acc0)
Jr = to_csr(ob.rowptr, ob.col, val_res)
Wm = (Jc - Jr).tocsr()                                       # all well terms
cells = np.asarray(wl.cells)
# A = reservoir + block-diagonal part of the well terms;  L = the rest (supported on perforated cells)
Wb = sp.bsr_matrix(Wm, blocksize=(3,3))
rows = np.repeat(np.arange(nc), np.diff(Wb.indptr)); isdiag = rows == Wb.indices
Dw = sp.bsr_matrix((Wb.data*isdiag[:,None,None], Wb.indices, Wb.indptr), shape=Wm.shape).tocsr()
A = (Jr + Dw).tocsr(); L = (Jc - A).tocsr()
print('n', 3*nc, 'nnz well terms', Wm.nnz, 'nnz L', L.nnz, 'rank L', np.linalg.matrix_rank(L.toarray()))
def ilu_of(mat):
    B = sp.bsr_matrix(mat, blocksize=(3,3)); B.sort_indices()
    # pattern must contain the diagonal; use stencil-only pattern for A (drop explicit zeros?)
    rp, cl, vl = B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.reshape(-1,9).copy()
    stt, lu = orc.ilu0(rp, cl, vl, position=position); assert stt == 0
    return lambda d: orc.ilu0_apply(rp, cl, lu, np.ascontiguousarray(d), position=position, relax=1.0)
# ilu0_apply takes eq-major? check by consistency below
def check(apply, mat, name):
    x = np.random.default_rng(0).standard_normal(3*nc); y = apply(mat @ x)
    print(name, 'ILU sanity |M^-1 A x - x|/|x|', np.linalg.norm(y-x)/np.linalg.norm(x))
# stencil-only pattern for A: rebuild with pattern of grid stencil
rp0, cl0 = orc.pattern(grid)
_, vres0, _, _ = orc.assemble(grid, tab, dt, ob.st, rp0, cl0, scale=tuple(ob.scale), accum0=ob.acc0)
A0 = to_csr(rp0, cl0, vres0) + Dw                            # same values, stencil pattern (+ diagonal well blocks)
assert abs(A0 - A).max() < 1e-9*abs(A).max()
MA = ilu_of(A0); MJ = ilu_of(Jc)
check(MA, A0, 'A'); check(MJ, Jc, 'Jclique')
# dense M for the exact Woodbury
I = np.eye(3*nc)
Minv = np.column_stack([MA(I[:,c]) for c in range(3*nc)])
Mmat = np.linalg.inv(Minv)
MLinv = np.linalg.inv(Mmat + L.toarray())
MW = lambda d: MLinv @ d
# crude variant: M^-1 restricted to block diagonal on the perforated rows
def gmres_its(prec, tol, two_stage=None):
    cnt = [0]
    def cb(rk): cnt[0] += 1
    Mop = spla.LinearOperator((3*nc,3*nc), matvec=prec)
    x, info = spla.gmres(Jc, b, M=Mop, rtol=tol, restart=40, maxiter=400, callback=cb, callback_type='pr_norm')
    return cnt[0], np.linalg.norm(b - Jc@x)/np.linalg.norm(b)
# CPR with an exact pressure solve (idealised first stage): weights = 1 per equation
wts = np.ones(3)
R = sp.kron(sp.eye(nc), wts.reshape(1,3)).tocsr()           # restriction: sum of equations
Cc = sp.kron(sp.eye(nc), np.array([[1.0],[0.0],[0.0]])).tocsr()   # pressure column
Ap = (R @ Jc @ Cc).tocsc(); Aplu = spla.splu(Ap)
def cpr(stage2):
    def f(d):
        xp = Aplu.solve(R @ d)
        x1 = Cc @ xp
        return x1 + stage2(d - Jc @ x1)
    return f
for tol in (1e-2, 1e-3, 1e-4):
    print('tol', tol)
    for name, pr in (('ILU0(A) [factored, now]', MA), ('exact Woodbury (M+L)^-1', MW), ('ILU0(J with cliques)', MJ)):
        print('   ILU only  %-28s its %3d  true red %.1e' % ((name,) + gmres_its(pr, tol)))
    for name, pr in (('ILU0(A) [factored, now]', MA), ('exact Woodbury (M+L)^-1', MW), ('ILU0(J with cliques)', MJ)):
        print('   CPR(exact p) + %-24s its %3d  true red %.1e' % ((name,) + gmres_its(cpr(pr), tol)))
