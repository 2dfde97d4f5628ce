import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import das_oracle as O
from qups_amd import ChannelData
rng = np.random.default_rng(4)
T, N, M = 200, 6, 6
fs = 20e6
x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
t0 = (1e-6 + 2.37e-7 * np.arange(M)).reshape(1, 1, M)
chd = ChannelData(torch.from_numpy(x), t0, fs)
r = chd.rectifyt0("cubic")
rref, t0ref = O.rectifyt0(x, t0, fs, "cubic")
y = r.data.cpu().numpy()
for m in range(M):
    d = np.abs(y[:, :, m] - rref[:, :, m]).max(axis=1)
    print(m, d.max(), np.argmax(d), np.nonzero(d > 1e-3)[0][:10])
