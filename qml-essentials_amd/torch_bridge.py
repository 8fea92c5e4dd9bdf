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

Scope: ``execution_type="expval"``; ``params`` of shape ``(layers, n_params)`` or a BATCH of
parameter sets ``(B_P, layers, n_params)`` (what the reference's loops over ``jax.grad`` accept,
``tests/test_model.py:1097-1145``: the result then has the model's ``(B_I, B_P, ...)`` batch axes and
every set receives its own gradient from ONE adjoint sweep over the whole batch), a batch of inputs
``(B_I, n_features)``, optional ``enc_params``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np


def _host_vjp(model, method, p, x, e, cot, n_out, wrt):
    """cot^T J for one parameter set on the host route: adjoint sweep, else the parameter-shift Jacobian."""
    from .adjoint import AdjointUnsupported

    B = cot.shape[0]
    try:
        if method != "adjoint":
            raise AdjointUnsupported
        v = model.gradient(params=p, inputs=x, enc_params=e, wrt=wrt, method="adjoint", cotangent=cot)
        return np.asarray(v).reshape(B, -1)
    except AdjointUnsupported:
        jac = np.asarray(model.gradient(params=p, inputs=x, enc_params=e, wrt=wrt))
        jac = jac.reshape(B, n_out, -1)
        return np.einsum("bk,bkj->bj", cot, jac)


def _host_vjps(model, method, p, x, e, cot, n_out, want_p, want_x, want_e):
    gp = _host_vjp(model, method, p, x, e, cot, n_out, "params").sum(axis=0) if want_p else None
    gx = _host_vjp(model, method, p, x, e, cot, n_out, "inputs") if want_x else None
    return gp, gx


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
            ctx.n_sets = int(p.shape[0]) if p.ndim == 3 else 1
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
            B_I = 1 if x is None else int(np.asarray(x).reshape(-1, model.n_input_feat).shape[0])
            B_P = ctx.n_sets
            rep_i, rep_p, _ = model.repeat_batch_axis
            cross = B_I > 1 and B_P > 1 and rep_i and rep_p
            B = B_I * B_P if cross else max(B_I, B_P)  # flattened batch, inputs slowest (model.py:1449-1481)

            grads = [None, None, None, None]
            if ctx.on_device and not (enc_params is not None and ctx.needs_input_grad[2]):
                # (a batch of parameter sets is one more leaf axis of the compiled call: CompiledCall.vjp
                # scatters every sample's angle gradient onto the rows of its own set)
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
            # ---- host route (noisy models, host arguments, parameter shift): one parameter set at a time
            g = grad_out.detach().cpu().numpy().astype(np.float64)
            if ctx.force_mean and n_out > 1:
                cot_all = np.repeat(g.reshape(B, 1) / n_out, n_out, axis=1)
            else:
                cot_all = g.reshape(B, n_out)
            if B_P > 1:
                gp = np.zeros(p.shape)
                gx = None if x is None else np.zeros((B_I, model.n_input_feat))
                for k in range(B_P):  # rows of set k: b_i * B_P + k (crossed) / row k (zipped)
                    rows = np.arange(B_I) * B_P + k if cross else (np.array([k]) if B_I in (1, B_P) else np.arange(B_I))
                    xk = x if (cross or x is None or B_I == 1) else np.asarray(x).reshape(B_I, -1)[k:k + 1]
                    one = _host_vjps(model, method, p[k], xk, e, cot_all[rows], n_out,
                                     ctx.needs_input_grad[0], gx is not None and ctx.needs_input_grad[1], False)
                    if one[0] is not None:
                        gp[k] = one[0].reshape(p.shape[1:])
                    if one[1] is not None:
                        if cross or B_I == 1:
                            gx += one[1].reshape(gx.shape)
                        else:
                            gx[k] = one[1].reshape(-1)
                if ctx.needs_input_grad[0]:
                    grads[0] = torch.as_tensor(gp, dtype=params.dtype, device=params.device)
                if gx is not None and ctx.needs_input_grad[1]:
                    grads[1] = torch.as_tensor(gx.reshape(tuple(inputs.shape)), dtype=inputs.dtype, device=inputs.device)
                if enc_params is not None and ctx.needs_input_grad[2]:
                    raise NotImplementedError("differentiable(): enc_params gradients need one parameter set per call")
                return tuple(grads)
            cot = cot_all

            def vjp(wrt):
                return _host_vjp(model, method, p, x, e, cot, n_out, wrt)

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
