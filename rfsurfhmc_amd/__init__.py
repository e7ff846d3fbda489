"""rfsurfhmc_amd -- MI355X-native misfit+gradient hot path of nqdu/RfSurfHmc.

csrc/            hand-written HIP kernels (gfx950) + the C ABI of include/rfsurf.h
model/lib/       ctypes drop-ins for the reference's pybind11 extensions libsurf / librf
model/           SurfWD, ReceiverFunc, Joint_RF_SWD: mirrors of the reference plugin API
pyhmc/           HamitonianMC, HMCDualAveraging: batched mirrors of the reference samplers
"""
__version__ = "0.1.0"
