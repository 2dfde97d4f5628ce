import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_gpu_configs import _setup,_run,_oracle_lattice
from tests.cases import rel_err
w,xc,prob=_setup('c3'); y,plan=_run(prob,xc)
img=y.cpu().numpy().reshape(w['I1'],w['I2'],order='F'); ref=_oracle_lattice(w,xc,32)
print('c3 lattice rel err', rel_err(img[::32,::32,None],ref))
