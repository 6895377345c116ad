"""iterative_inference_segm_amd -- MI355X-native iterative-inference hot path.

FCN-8 forward -> conditional DAE -> N refinement steps on the softmax map, behind the
reference's own entry point and knobs (iterative_inference.py: segmentation_net, dae_dict,
step, num_iter).  All arithmetic runs in hand-written HIP kernels for gfx950 (csrc/, C ABI in
include/iiseg.h); importing the compute modules fails loudly if libiiseg_hip.so is missing.
"""
__version__ = '0.1.0'
