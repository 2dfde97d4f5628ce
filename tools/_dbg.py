import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from tests.cases import make_case, rel_err
from tests.test_gpu_parity import run_das, run_oracle
for interp in ['nearest','linear','lanczos3']:
    case = make_case(seq='FSA', interp=interp, seed=2, I1=150, I2=19)
    ref = run_oracle(case); out, plan = run_das(case, kernel=2)
    e=np.abs(out-ref)[...,0,0]
    print(interp, rel_err(out,ref), plan.fallback_tiles(), 'bad cols', np.where(e.max(axis=0)[:,0]>1e-3*np.abs(ref).max())[0], 'bad rows', np.where(e.max(axis=1)[:,0]>1e-3*np.abs(ref).max())[0][:20])
