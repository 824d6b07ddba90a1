"""CPU: which stage decides how far the golden case m64p2 (16 peaks in a noise-free m = 64 Hankel matrix: 48 of the 64 retained
singular values are rounding noise, every line carries eps * 1e8 of it) lands from the reference's output?  The pipeline is
restated in numpy with the SVD and the eigen-solver swapped independently between LAPACK (zgesdd / zgesvd, zgeev) and this
repository's algorithms (tests/hostsim: the templates the kernels instantiate).  Diagnostic (uses the oracle and hostsim:
test infrastructure).  Result: the eigen-solver moves the lines by 1e-11, the SVD by 3e-9 ... 9e-9 -
LAPACK's own two drivers included."""
import ctypes, numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import canonical, keep_mask
from oracle import kbdm_oracle as O
import scipy.linalg as sla
P=ctypes.c_void_p
hs=ctypes.CDLL(os.path.join(ROOT, 'tests', 'hostsim', '_build', 'libhostsim.so'))
g=dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'kbdm_golden.npz'),allow_pickle=True))
name='m64p2'
m,l,p=(int(x) for x in g[f'{name}__meta']); q=float(g[f'{name}__q'][0])
sig=g[str(g[f'{name}__sig'])]
want=g[f'{name}__kept']
dwell=5e-4
def lines_from(L,s,R,eigfun,U0,Up):
    dsqi=1/np.sqrt(s)
    W=(dsqi[:,None]*(L.conj().T@Up@R))*dsqi[None,:]
    mu,Pm=eigfun(W)
    B=R@(dsqi[:,None]*Pm)
    N=np.einsum('ik,ik->k',B,U0@B)
    B=B*np.sqrt(1/N)
    Dsq=sig[:m]@B
    D=Dsq**2
    A=np.abs(D); PH=np.angle(D); T2=-dwell/np.log(np.abs(mu)); F=np.angle(mu)/(2*np.pi*dwell)
    ll=np.column_stack([A,T2,F,PH])
    return canonical(ll[keep_mask(ll)])
def dist(k):
    if k.shape!=want.shape: return 'shape',k.shape
    rel=[float((np.abs(k[:,c]-want[:,c])/np.abs(want[:,c])).max()) for c in range(3)]
    return ['%.2e'%x for x in rel]+['%.2e'%float(np.abs(np.angle(np.exp(1j*(k[:,3]-want[:,3])))).max())]
U0,Up1,Up=O.compute_U_matrices(sig,m,p)
def np_svd(A,drv='gesdd'):
    L,s,Rh=sla.svd(A,lapack_driver=drv); return L,s,Rh.conj().T
def hs_svd(A):
    mm=A.shape[0]; Af=np.asfortranarray(A); L=np.zeros((mm,mm),complex,order='F'); R=np.zeros((mm,mm),complex,order='F'); s=np.zeros(mm)
    hs.hs_svd(Af.ctypes.data_as(P),mm,L.ctypes.data_as(P),s.ctypes.data_as(P),R.ctypes.data_as(P)); return L,s,R
def np_eig(W): return sla.eig(W)
def hs_eig(W):
    n=W.shape[0]; Wf=np.asfortranarray(W); mu=np.zeros(n,complex); Pm=np.zeros((n,n),complex,order='F')
    hs.hs_eig(Wf.ctypes.data_as(P),n,mu.ctypes.data_as(P),Pm.ctypes.data_as(P)); return mu,Pm
for sn,sf in (('gesdd',lambda A:np_svd(A,'gesdd')),('gesvd',lambda A:np_svd(A,'gesvd')),('ours',hs_svd)):
    L,s,R=sf(Up1)
    for en,ef in (('zgeev',np_eig),('ours',hs_eig)):
        print(sn,en,dist(lines_from(L,s,R,ef,U0,Up)))
