import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from inverse_compositional_algorithm_amd.engine import IcaEngine
from inverse_compositional_algorithm_amd import synthetic as syn
from oracle import ica_oracle as oc
eng = IcaEngine(0)
H_, W_ = 160, 200
for t, pv in [(4, [-6.0, 2.0, 0.05, 0.3, -0.25, 0.02]), (4, [0,0,0,0,0,0]), (3, [5.0, 4.0, 0.45, 0.1])]:
    n = oc.nparams(t)
    prs = [syn.make_pair(777+17*4+i, H_, W_, 1, 4) for i in range(2)]
    d1 = torch.from_numpy(np.stack([np.moveaxis(p[0],2,0) for p in prs])).to(eng.device); d2 = torch.from_numpy(np.stack([np.moveaxis(p[1],2,0) for p in prs])).to(eng.device)
    prob = eng.problem(2, H_, W_, 1, t, 0, delta=5)
    pp = np.zeros((2,8)); pp[0,:n] = pv; pp[1,:n] = np.array(pv)*0.5
    Hm, bv = eng.normal_equations(d1, d2, torch.from_numpy(pp), prob)
    Hm = Hm.cpu().numpy()
    for i in range(2):
        Ix, Iy = oc.gradient(prs[i][0].astype(np.float64), True, 5)
        DIJ = oc.steepest_descent_images(Ix, Iy, oc.jacobian(t, W_, H_), n)
        Hw = oc.hessian(DIJ)
        print(t, i, 'max rel err', np.max(np.abs(Hm[i,:n,:n]-Hw)/np.sqrt(np.outer(np.diag(Hw),np.diag(Hw)))))
