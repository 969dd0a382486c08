import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, piqp_amd as hip
from oracle import pyorc as orc
from qp_io import load_qp
name=sys.argv[1] if len(sys.argv)>1 else 'qp_robot_arm_sqp'
q=load_qp(name)
args=(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
so=orc.Solver(); so.settings.kkt_solver=1
so.setup(*args, sparse=True)
states=so.record_states()
so.solve()
od=so.data()   # scaled data
# build hip data from the SCALED oracle data so both see identical matrices
Pu=od.csc('P_utri'); AT=od.csc('AT'); GT=od.csc('GT')
n,p,m=od.n,od.p,od.m
class D(hip.SparseData):
    def __init__(self):
        self.n,self.p,self.m=n,p,m
        self.P_utri=Pu; self.AT=AT; self.GT=GT
        self.h_l_idx=od.idx('h_l'); self.h_u_idx=od.idx('h_u'); self.x_l_idx=od.idx('x_l'); self.x_u_idx=od.idx('x_u')
        self.n_h_l,self.n_h_u,self.n_x_l,self.n_x_u=od.counts()
        self.x_b_scaling=od.vec('x_b_scaling').copy()
d=D()
kh=hip.KKTSystem(d, hip.default_settings(kkt_solver=1))
ko=orc.KKTSystem(od, orc.Settings(kkt_solver=1))
fs=[s for s in states if s['kind']==0]
ss=[s for s in states if s['kind']==1]
Pf=(Pu+sp.triu(Pu,1).T).tocsr(); A=AT.T.tocsr(); G=GT.T.tocsr()
for it in [0,3,6,8,9,10,12,20,40]:
    if it>=len(fs): break
    st=fs[it]; rhs=ss[min(2*it+1,len(ss)-1)]['vars']
    okh=kh.update_scalings_and_factor(False, st['rho'], st['delta'], st['vars'])
    oko=ko.update_scalings_and_factor(False, st['rho'], st['delta'], st['vars'])
    _,lh=kh.solve(rhs); _,lo=ko.solve(rhs)
    xr,zr=ko.x_reg(),ko.z_reg(); rx,rz=ko.rhs_x_bar(),ko.rhs_z_bar(); ry=rhs['y']
    def resid(l):
        z=l['z_u']-l['z_l']
        L=np.longdouble
        r1=rx.astype(L)-(Pf@l['x']).astype(L)-xr.astype(L)*l['x']-(AT@l['y']).astype(L)-(GT@z).astype(L)
        r2=ry.astype(L)-(A@l['x']).astype(L)+L(st['delta'])*l['y']
        r3=rz.astype(L)-(G@l['x']).astype(L)+zr.astype(L)*z
        return float(max(np.abs(r1).max(), np.abs(r2).max() if p else 0, np.abs(r3).max() if m else 0))
    nrm=max(np.abs(rx).max(), np.abs(ry).max() if p else 0, np.abs(rz).max() if m else 0)
    dx=np.abs(lh['x']-lo['x']).max()/np.abs(lo['x']).max()
    print(f"it {it:3d} rho {st['rho']:.1e} delta {st['delta']:.1e} ok {okh},{oko} relres hip {resid(lh)/nrm:.2e} orc {resid(lo)/nrm:.2e} rel dx {dx:.2e}")
kh.backend().print_info()
