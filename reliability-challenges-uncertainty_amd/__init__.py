"""MI355X-native MC-dropout / ensemble uncertainty path (HIP kernels behind a C ABI).

Host-side mirror of the reference's three seams:
  model   -> rcu_amd.model.UNet
  steps   -> rcu_amd.steps (McPredictStep, MultiPredictionSummary, EnsemblePredictionStep, ...)
  metrics -> rcu_amd.evaluation (EceBinaryNumpy, UncertaintyAndCorrectionEvalNumpy, ...)
Everything computes through librcu_hip.so (include/rcu.h); there is no CPU fallback.
"""
__version__ = '0.1.0'
