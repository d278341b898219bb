"""MI355X-native discrete graph-DDPM hot path (PPGN denoiser, edge-flip forward process, reverse sampling).

Drop-in for the corresponding pieces of kilianhae/discrete_DPPM_Graphs; all device work goes through
libdppm_hip.so (include/dppm_hip.h).  Importing `discrete_dppm_graphs_amd.ops` (or anything that needs a kernel)
fails loudly when the HIP extension has not been built."""
__version__ = "0.1.0"
