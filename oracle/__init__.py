"""CPU oracle for the discrete graph-DDPM hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithms of the reference path (kilianhae/discrete_DPPM_Graphs)
so that the HIP kernels can be checked on machines where the reference itself is absent.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product package `discrete_dppm_graphs_amd` never does (tests/test_boundary.py greps for that).

Parity status: PINNED.  Every function here is checked in tests/test_oracle_golden.py against
fixtures in tests/golden/*.npz that were produced by running the reference's own code
(tests/golden/gen_golden.py, executed in the build container against /root/reference).

Modules
  philox      counter-based Philox4x32-10 RNG (new design; the reference uses torch's global RNG)
  diffusion   forward edge-flip q(A_t|A_0), reverse steps (simple / VLB), x_T init, masks,
              BCE / VLB losses, sigma_lin      [utils/graph_utils.py, sample_ppgn_*.py, ppgn_*.py]
  ppgn        the PPGN ("Powerful") denoiser forward as a functional torch-CPU restatement with
              autograd for gradients             [model/ppgn.py, util/model_helper.py]
"""
