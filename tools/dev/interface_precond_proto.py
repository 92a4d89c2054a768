#!/usr/bin/env python3
"""CPU prototype (scipy on the oracle-assembled system of a coupled strip): what would a better treatment of the interface unknowns in the
block-local preconditioner buy?  Interior unknowns of every block solved EXACTLY (what the multigrid cycle approximates), BiCGStab
iterations to rtol 1e-10 with
  jacobi     interior solve + identity on the perimeter rows                     (today's preconditioner, idealised)
  lower      e_p = f_p, then e_I = A_II^-1 (f_I - A_Ip e_p)                      (perimeter values as Dirichlet data of the interior solve)
  upper      interior solve, then e_p = f_p - A_pI e_I                          (one perimeter-row application behind the cycle, nothing in front)
  symmetric  lower, then e_p = f_p - A_pI e_I                                     (+ one perimeter-row application)
usage: interface_precond_proto.py [blocks = 2] [ni = 65] [nj = 65]"""
import os, sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from tests.conftest import OracleMesh, oracle_tfi
from turbomesh_amd import configs
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 65
nj = int(sys.argv[3]) if len(sys.argv) > 3 else ni
mesh = configs.strip(k, ni, nj, tfi=oracle_tfi) if k > 1 else configs.single_block(ni, nj, tfi=oracle_tfi)
rng = np.random.default_rng(5)
for b in mesh.blocks:
    d = b.points.data
    d[1:-1, 1:-1] += 0.25 / d.shape[0] * (rng.random(d[1:-1, 1:-1].shape) - 0.5)
om = OracleMesh(mesh)
s = oracle.System(om); s.fill(0); s.fill_x_specific()
A = s.csr().tocsr(); b = s.rhs_x.copy(); n = A.shape[0]
d = A.diagonal(); B = (sp.diags(1.0 / d) @ A).tocsr(); bs = b / d
x0 = om.flat()[:, 0].copy()
if os.environ.get('PROTO_RANDOM'):   # a generic right-hand side instead of the Picard step's (whose perimeter residual is special)
    bs = B @ np.random.default_rng(7).standard_normal(n); x0 = np.zeros(n)
interior = np.zeros(n, bool)
off = 0
for bl in om.blocks:
    a, c = bl.shape[:2]
    m = np.zeros((a, c), bool); m[1:-1, 1:-1] = True
    interior[off:off + a * c] = m.ravel(); off += a * c
I, P = np.where(interior)[0], np.where(~interior)[0]
BII = spla.splu(B[I][:, I].tocsc()); BIP = B[I][:, P].tocsr(); BPI = B[P][:, I].tocsr()
def jacobi(f):
    e = f.copy(); e[I] = BII.solve(f[I]); return e
def lower(f):
    e = f.copy(); e[I] = BII.solve(f[I] - BIP @ f[P]); return e
def upper(f):
    e = jacobi(f); e[P] = f[P] - BPI @ e[I]; return e
def symmetric(f):
    e = lower(f); e[P] = f[P] - BPI @ e[I]; return e
BPP = spla.splu(B[P][:, P].tocsc())
def symmetric_pp(f):   # the same with the perimeter-perimeter couplings (along the interface, connected copies) solved exactly instead of ignored
    e = f.copy(); e[P] = BPP.solve(f[P]); e[I] = BII.solve(f[I] - BIP @ e[P]); e[P] = BPP.solve(f[P] - BPI @ e[I]); return e
BPPo = (B[P][:, P] - sp.identity(len(P))).tocsr()   # off-diagonal part (unit diagonal: the rows are equilibrated)
def jac(rhs, k):   # k Jacobi sweeps on B_PP e = rhs from e = rhs
    e = rhs.copy()
    for _ in range(k): e = rhs - BPPo @ e
    return e
def make_sym_jac(k1, k2):
    def f_(f):
        e = f.copy(); e[P] = jac(f[P], k1); e[I] = BII.solve(f[I] - BIP @ e[P]); e[P] = jac(f[P] - BPI @ e[I], k2); return e
    return f_
def bicgstab(prec, rtol=1e-10, maxit=20000):
    x = x0.copy(); r = bs - B @ x; rh = r.copy(); rho = alpha = om_ = 1.0; v = np.zeros(n); p = np.zeros(n)
    tol = rtol * np.linalg.norm(bs)
    for it in range(1, maxit + 1):
        rho_new = rh @ r; beta = (rho_new / rho) * (alpha / om_); p = r + beta * (p - om_ * v)
        ph = prec(p); v = B @ ph; alpha = rho_new / (rh @ v); s_ = r - alpha * v
        sh = prec(s_); t = B @ sh; om_ = (t @ s_) / (t @ t); x += alpha * ph + om_ * sh; r = s_ - om_ * t; rho = rho_new
        if np.linalg.norm(r) <= tol: return it
    return maxit
print(f"{k} x {ni} x {nj}: n {n}, perimeter unknowns {len(P)}")
for name, f in (("diagonal only", lambda v: v), ("jacobi", jacobi), ("lower", lower), ("upper", upper), ("symmetric", symmetric), ("symmetric + B_PP", symmetric_pp), ("sym, Jacobi 0+1", make_sym_jac(0, 1)), ("sym, Jacobi 0+2", make_sym_jac(0, 2)), ("sym, Jacobi 0+4", make_sym_jac(0, 4)), ("sym, Jacobi 2+2", make_sym_jac(2, 2)), ("sym, Jacobi 4+4", make_sym_jac(4, 4))):
    print(f"  {name:14s} {bicgstab(f):6d} iterations", flush=True)
if os.environ.get("PROTO_CHECK"):
    # one application of each preconditioner as a stationary correction: how much of the residual does it remove?
    r0 = bs - B @ x0
    for name, f in (("jacobi", jacobi), ("lower", lower), ("upper", upper), ("symmetric", symmetric)):
        x1 = x0 + f(r0)
        print(f"  one correction with {name:10s}: |r| {np.linalg.norm(r0):.3e} -> {np.linalg.norm(bs - B @ x1):.3e}   (nnz B_PI {BPI.nnz}, B_IP {BIP.nnz}, |B_PP - I| {abs(B[P][:, P] - sp.identity(len(P))).max():.2e})")
    f = np.random.default_rng(1).standard_normal(n)
    for name, g in (("jacobi", jacobi), ("lower", lower), ("upper", upper), ("symmetric", symmetric)):
        print(f"  random f: |B M^-1 f - f| / |f| with {name:10s}: {np.linalg.norm(B @ g(f) - f) / np.linalg.norm(f):.3e}")
    kinds = s.kinds() if hasattr(s, "kinds") else None
    print("  rows of B_IP with non-zeros:", len(np.unique(BIP.nonzero()[0])), " columns:", len(np.unique(BIP.nonzero()[1])), " B_PI rows:", len(np.unique(BPI.nonzero()[0])))
