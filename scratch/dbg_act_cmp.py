import numpy as np
R0 = np.load("gpurun_out/R_nows.npy"); R1 = np.load("gpurun_out/R_ws.npy")
a0 = np.load("gpurun_out/act_nows.npy"); a1 = np.load("gpurun_out/act_ws.npy")
print("R diff", np.abs(R0 - R1).max(), "act diff", np.abs(a0 - a1).max())
flip = (a0 > 0) != (a1 > 0)
print("sign flips", flip.sum(), "of", flip.size)
idx = np.argwhere(flip)
for i in idx[:20]:
    print(i, a0[tuple(i)], a1[tuple(i)])
d = np.abs(a0 - a1); j = np.unravel_index(d.argmax(), d.shape); print("max diff at", j, a0[j], a1[j])
