"""PyTorch autograd bridge: ``model(...)`` as a differentiable torch function.

The reference is trained with ``jax.grad`` through ``Model.__call__``
(``tests/test_model.py:20-70,1097-1145,1297-1333``, ``docs/training.md``).  Here the same
workflow runs on PyTorch's autograd: the forward pass is the HIP engine, the backward pass is
ONE adjoint sweep (:mod:`adjoint`, ``qmle_adjoint_gradient``) per differentiated argument, with
the incoming ``grad_output`` as cotangent -- so any torch cost function and any
``torch.optim`` optimiser work unchanged:

    f = differentiable(model)
    params = torch.tensor(model.params[0], device="cuda", requires_grad=True)
    loss = ((f(params, x, force_mean=True) - y) ** 2).mean()
    loss.backward()            # params.grad by adjoint differentiation

Scope: ``execution_type="expval"``, one parameter set per call (``params`` of shape
``(layers, n_params)``), a batch of inputs ``(B, n_features)``, optional ``enc_params``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np


def differentiable(model, method: str = "adjoint"):
    """Returns ``f(params, inputs=None, enc_params=None, force_mean=False) -> torch.Tensor``
    (CUDA, float32) that takes part in autograd.  ``method``: ``"adjoint"`` (default; falls
    back to the parameter-shift rule where the adjoint sweep has no rule) or
    ``"parameter-shift"``."""
    import torch

    from .adjoint import AdjointUnsupported

    def _np(t):
        return None if t is None else (t.detach().cpu().numpy().astype(np.float64)
                                       if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float64))

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, params, inputs, enc_params, force_mean):
            ctx.save_for_backward(params, inputs if isinstance(inputs, torch.Tensor) else None,
                                  enc_params if isinstance(enc_params, torch.Tensor) else None)
            ctx.force_mean = bool(force_mean)
            # CUDA tensors in, nothing noisy: forward AND backward stay on the GPU
            ctx.on_device = (method == "adjoint" and params.is_cuda and model.noise_params is None
                             and model.shots is None
                             and (inputs is None or (isinstance(inputs, torch.Tensor) and inputs.is_cuda)))
            p, x, e = _np(params), _np(inputs), _np(enc_params)
            if p.ndim == 3 and p.shape[0] != 1:
                raise NotImplementedError("differentiable(): one parameter set per call")
            ctx.host = (p, x, e)
            if ctx.on_device:
                out = model(params=params.detach(), inputs=None if inputs is None else inputs.detach(),
                            enc_params=e, execution_type="expval", force_mean=force_mean)
                if isinstance(out, torch.Tensor):
                    ctx.n_out = int(model._result_shape[0])
                    return out.to(torch.float32)
                ctx.on_device = False
            out = np.asarray(model(params=p, inputs=x, enc_params=e, execution_type="expval",
                                   force_mean=force_mean))
            ctx.n_out = int(model._result_shape[0])
            return torch.as_tensor(out, dtype=torch.float32, device="cuda")

        @staticmethod
        def backward(ctx, grad_out):
            params, inputs, enc_params = ctx.saved_tensors
            p, x, e = ctx.host
            n_out = ctx.n_out
            B = 1 if x is None else int(np.asarray(x).reshape(-1, model.n_input_feat).shape[0])
            g = grad_out.detach().cpu().numpy().astype(np.float64)
            if ctx.force_mean and n_out > 1:
                cot = np.repeat(g.reshape(B, 1) / n_out, n_out, axis=1)
            else:
                cot = g.reshape(B, n_out)

            def vjp(wrt):
                try:
                    if method != "adjoint":
                        raise AdjointUnsupported
                    v = model.gradient(params=p, inputs=x, enc_params=e, wrt=wrt, method="adjoint",
                                       cotangent=cot)
                    return np.asarray(v).reshape(B, -1)
                except AdjointUnsupported:
                    jac = np.asarray(model.gradient(params=p, inputs=x, enc_params=e, wrt=wrt))
                    jac = jac.reshape(B, n_out, -1)
                    return np.einsum("bk,bkj->bj", cot, jac)

            grads = [None, None, None, None]
            if ctx.on_device and not (enc_params is not None and ctx.needs_input_grad[2]):
                try:
                    gp, gx = model.vjp_device(
                        params.detach(), None if inputs is None else inputs.detach(),
                        grad_out.reshape(B, -1) if not ctx.force_mean else grad_out.reshape(B),
                        enc_params=e, force_mean=ctx.force_mean)
                    grads[0] = gp if ctx.needs_input_grad[0] else None
                    grads[1] = gx if (inputs is not None and ctx.needs_input_grad[1]) else None
                    return tuple(grads)
                except (NotImplementedError, AdjointUnsupported):
                    pass
            if ctx.needs_input_grad[0]:
                grads[0] = torch.as_tensor(vjp("params").sum(axis=0).reshape(tuple(params.shape)),
                                           dtype=params.dtype, device=params.device)
            if inputs is not None and ctx.needs_input_grad[1]:
                grads[1] = torch.as_tensor(vjp("inputs").reshape(tuple(inputs.shape)),
                                           dtype=inputs.dtype, device=inputs.device)
            if enc_params is not None and ctx.needs_input_grad[2]:
                grads[2] = torch.as_tensor(vjp("enc_params").sum(axis=0).reshape(tuple(enc_params.shape)),
                                           dtype=enc_params.dtype, device=enc_params.device)
            return tuple(grads)

    def f(params, inputs=None, enc_params=None, force_mean: bool = False):
        return _Fn.apply(params, inputs, enc_params, force_mean)

    return f
