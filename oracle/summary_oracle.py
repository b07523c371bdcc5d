"""Oracle: MC / ensemble aggregation of T stacked probability maps.  TEST INFRASTRUCTURE ONLY.

Follows rechun/dl/customsteps.py:16-71 and common/utils/torchhelper.py:53-54 of the reference.
"""
import torch
import torch.nn.functional as F


def torch_entropy(p, dim=-1, keepdim=False):
    """Natural-log entropy with 0*log 0 := 0 (torchhelper.py:53-54)."""
    return -torch.where(p > 0, p * p.log(), torch.zeros((), dtype=p.dtype)).sum(dim=dim, keepdim=keepdim)


def mc_probabilities(forward_fn, x, mask_sets):
    """Weight-scaling pass (dropout off) followed by one stochastic pass per mask set, softmax over
    the class dim each (customsteps.py:22-36).  forward_fn(x, masks_or_None) -> logits."""
    ws = F.softmax(forward_fn(x, None), 1)
    multi = torch.stack([F.softmax(forward_fn(x, m), 1) for m in mask_sets])
    return ws, multi


def ensemble_probabilities(forward_fns, x):
    """K eval-mode members, softmax each, stacked (bin-dl/brats_test_ensemble.py:84-94)."""
    return torch.stack([F.softmax(f(x, None), 1) for f in forward_fns])


def multi_prediction_summary(multi, do_mi=False, do_var=False):
    """multi: [T, N, C, H, W] -> dict with the reference's output keys (customsteps.py:57-71):
    probabilities [N,C,H,W] = mean over T; entropy [N,1,H,W] of the mean; mutual_info = entropy
    minus the mean per-sample entropy; variance = unbiased var over T, mean over C."""
    multi = torch.as_tensor(multi)
    out = {}
    probs = multi.mean(dim=0)
    out['probabilities'] = probs
    ent = torch_entropy(probs, dim=1, keepdim=True)
    out['entropy'] = ent
    if do_mi:
        out['mutual_info'] = ent - torch_entropy(multi, dim=2, keepdim=True).mean(dim=0)
    if do_var:
        out['variance'] = multi.var(dim=0).mean(dim=1, keepdim=True)
    return out


def aleatoric_outputs(logits, sigma_raw, is_log_sigma=False):
    """bin-dl/brats_test_aleatoric.py:63-73: sigma = exp(raw) or |raw|; softmax of the mean logits."""
    sigma = sigma_raw.exp() if is_log_sigma else sigma_raw.abs()
    return dict(logits=logits, sigma=sigma, probabilities=F.softmax(logits, 1))
