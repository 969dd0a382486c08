import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, piqp_amd as hip
from oracle import pyorc as orc
from qp_io import load_qp
q=load_qp(sys.argv[1] if len(sys.argv)>1 else 'qp_robot_arm_sqp')
args=(q["P"], q["c"], q["A"], q["b"], q["G"], q["h_l"], q["h_u"], q["x_l"], q["x_u"])
sh=hip.SparseSolver(); sh.settings.kkt_solver=1; sh.enable_trace()
so=orc.Solver(); so.settings.kkt_solver=1; so.enable_trace()
sh.setup(*args); so.setup(*args, sparse=True)
print(sh.solve(), so.solve(), sh.info.iter, so.info.iter, sh.info.n_factor, so.info.n_factor)
th,to=sh.trace(),so.trace()
np.set_printoptions(linewidth=250, precision=4)
for i in range(min(len(th),len(to),40)):
    print(i, th[i][[1,4,5,6,7,8]], '|', to[i][[1,4,5,6,7,8]])
print('--- tail hip')
for i in list(range(40,len(th),15))+[len(th)-1]:
    print(i, th[i][[1,2,3,4,5,6,7,8,9,10]])
print('--- tail orc')
for i in range(40,len(to),6):
    print(i, to[i][[1,2,3,4,5,6,7,8,9,10]])
