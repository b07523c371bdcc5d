"""MI355X-native MC-dropout / ensemble uncertainty path (HIP kernels behind a C ABI).

Host-side mirror of the reference's three seams:
  model   -> rcu_amd.model.UNet
  steps   -> rcu_amd.steps (McPredictStep, MultiPredictionSummary, EnsemblePredictionStep, ...)
  metrics -> rcu_amd.evaluation (EceBinaryNumpy, UncertaintyAndCorrectionEvalNumpy, ...)
Everything computes through librcu_hip.so (include/rcu.h); there is no CPU fallback.
"""
import os as _os_env

# Kernel arguments in device memory: the HIP runtime then does not pull them over PCIe at every launch.  A forward pass is 23 dependent
# launches and the GPU idles between them for about the launch latency (measured on the headline bench: 149.5 -> 150.8
# MC-sample-volumes/s).  Read by the runtime when it initialises, so it has to be in the environment before the first HIP call; a
# value the user has set wins.
_os_env.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
# HIP streams share the runtime's hardware queues (4 by default) and a queue runs in order: with the compute stream, the second stream
# lane, the copy stream of the input prefetch, the download stream and the subject-step stream of the test loop, a small download or
# metric kernel landed in the queue of the compute stream and waited for a whole volume's forward passes (measured in
# bin-dl/brats_test_default.py: 0.19 s per subject step with 4 queues, 0.01-0.04 s with 8).  Same rule: before the first HIP call.
_os_env.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
