"""``ExactGP`` and ``DefaultPredictionStrategy`` -- train / prior / posterior dispatch and the
prediction caches, following ``gpytorch/models/exact_gp.py:265-333,355-430`` and
``gpytorch/models/exact_prediction_strategies.py:46-92,267-321,331-478``:

  * mean cache  = K_hat^-1 (y - mu)  by preconditioned mBCG at ``eval_cg_tolerance``   (:278-286, exact_gp.py:324)
  * mean        = K_*X @ mean_cache + mu_*   (rectangular fused K*V, t = 1)             (:396,411)
  * covariance  = K_** - (K_*X S)(K_*X S)^T with S the Lanczos root-inverse (LOVE,
                  ``fast_pred_var``)                                                      (:267-272,464-478)
                  or K_** - K_*X K_hat^-1 K_X* by an n_test-column mBCG                  (:431-462)
"""
from __future__ import annotations

import warnings

import torch

from . import settings
from .module import Module
from .operators import (
    DenseLinearOperator,
    LinearOperator,
    MatmulLinearOperator,
    SumLinearOperator,
    ZeroLinearOperator,
    to_dense,
    to_linear_operator,
)


class GPInputWarning(UserWarning):
    pass


class DefaultPredictionStrategy:
    def __init__(self, train_inputs, train_prior_dist, train_labels, likelihood, root=None, inv_root=None):
        self._train_shape = train_prior_dist.event_shape
        self.train_inputs = train_inputs
        self.train_prior_dist = train_prior_dist
        self.train_labels = train_labels.reshape(*train_labels.shape[: -len(self._train_shape)], self._train_shape.numel())
        self.likelihood = likelihood
        self._last_test_train_covar = None
        mvn = self.likelihood(train_prior_dist, train_inputs)
        self.lik_train_train_covar = mvn.lazy_covariance_matrix
        self._mean_cache = None
        self._masked_cache = None
        self._covar_cache = None
        self._inv_root = inv_root

    @property
    def num_train(self):
        return self._train_shape.numel()

    def _masked_mean_cache(self, policy):
        """exact_prediction_strategies.py:287-315: the training targets hold NaNs and the policy is "mask" or "fill".  Both solve the system of the
        observed points only ("fill" zeroes the missing rows and columns of a dense matrix and keeps its diagonal -- the observed block of that
        solve IS the masked solve) and mark the missing entries of the cache NaN; a point counts as observed if every batch member observes it."""
        mvn = self.likelihood(self.train_prior_dist, self.train_inputs)
        train_mean, op = mvn.loc, mvn.lazy_covariance_matrix.evaluate_kernel()
        n = self.train_labels.shape[-1]
        idx = (~torch.isnan(self.train_labels.reshape(-1, n)).any(dim=0)).nonzero().squeeze(-1)
        sub = op.restrict(idx) if hasattr(op, "restrict") else op[idx][:, idx]
        offset = (self.train_labels - train_mean)[..., idx].unsqueeze(-1)
        mc = torch.full_like(self.train_labels, float("nan"))
        mc[..., idx] = sub.solve(offset).squeeze(-1)
        return mc.detach() if settings.detach_test_caches.on() else mc

    @property
    def mean_cache(self):
        """exact_prediction_strategies.py:275-321: one cache per observation-NaN policy."""
        policy = settings.observation_nan_policy.value()
        if policy != "ignore" and bool(torch.isnan(self.train_labels).any()):
            if policy == "fill":
                warnings.warn("Observation NaN policy 'fill' makes the kernel matrix dense during exact prediction.", RuntimeWarning)
            if self._masked_cache is None:
                self._masked_cache = self._masked_mean_cache(policy)
            return self._masked_cache
        if self._mean_cache is None:
            mvn = self.likelihood(self.train_prior_dist, self.train_inputs)
            train_mean, train_train_covar = mvn.loc, mvn.lazy_covariance_matrix
            offset = (self.train_labels - train_mean).unsqueeze(-1)
            op = train_train_covar.evaluate_kernel()
            if (settings.fast_pred_var.on() and self._covar_cache is None and self._inv_root is None and offset.dim() == 2
                    and hasattr(op, "can_fuse_caches") and op.can_fuse_caches()):
                # LOVE is on and both caches are missing: one sequence of two-column products serves the mean-cache CG and the
                # Lanczos run together (FusedKernelAddedDiagLinearOperator.solve_and_root_inv)
                mc, root = (op.detach() if settings.detach_test_caches.on() else op).solve_and_root_inv(offset)
                mc = mc.squeeze(-1)
                self._covar_cache = root.root.detach() if settings.detach_test_caches.on() else root.root
            else:
                mc = op.solve(offset).squeeze(-1)
            if settings.detach_test_caches.on():
                mc = mc.detach()
            self._mean_cache = mc
        return self._mean_cache

    @property
    def covar_cache(self):
        """exact_prediction_strategies.py:267-272: root of K_hat^-1 (n x m)."""
        if self._covar_cache is None:
            cov = self.lik_train_train_covar
            if settings.detach_test_caches.on():
                cov = cov.detach()
            root = self._inv_root if self._inv_root is not None else to_dense(cov.root_inv_decomposition().root)
            self._covar_cache = root.detach() if settings.detach_test_caches.on() else root
        return self._covar_cache

    def exact_prediction(self, test_mean, test_test_covar, test_train_covar):
        """:331-369."""
        if sum(test_train_covar.shape[-2:]) <= settings.max_eager_kernel_size.value():
            test_train_covar = to_dense(test_train_covar)
            test_test_covar = to_dense(test_test_covar)
        return (
            self.exact_predictive_mean(test_mean, test_train_covar),
            self.exact_predictive_covar(test_test_covar, test_train_covar),
        )

    def exact_predictive_mean(self, test_mean, test_train_covar):
        """:371-412."""
        mc = self.mean_cache
        if settings.observation_nan_policy.value() != "ignore":
            mc = torch.nan_to_num(mc, nan=0.0)      # (:397-410: the columns of the missing observations drop out of the product)
        res = (test_train_covar @ mc.unsqueeze(-1)).squeeze(-1)
        return res + test_mean

    def exact_predictive_covar(self, test_test_covar, test_train_covar):
        """:414-478."""
        if settings.fast_pred_var.on():
            self._last_test_train_covar = test_train_covar
        if settings.skip_posterior_variances.on():
            return ZeroLinearOperator(*test_test_covar.shape, dtype=test_train_covar.dtype, device=test_train_covar.device)
        if settings.fast_pred_var.off():
            dist = self.train_prior_dist.__class__(torch.zeros_like(self.train_prior_dist.mean), self.train_prior_dist.lazy_covariance_matrix)
            train_train_covar = self.likelihood(dist, self.train_inputs).lazy_covariance_matrix
            if settings.detach_test_caches.on():
                train_train_covar = train_train_covar.detach()
            ttc = to_dense(test_train_covar)
            rhs_in = ttc.mT.contiguous()
            if rhs_in.dim() > 2 and not train_train_covar.batch_shape:  # batch of test sets against one training set
                from .operators import BatchLinearOperator

                train_train_covar = BatchLinearOperator.replicate(train_train_covar, rhs_in.shape[:-2])
            if (settings.rhs_refinement.on() and ttc.dtype == torch.float32 and rhs_in.dim() == 2 and rhs_in.shape[-1] <= 4096
                    and hasattr(train_train_covar, "float64_product_available")):
                # K_** - K_*X K_hat^-1 K_X* is 1 - 0.9998.. at a well-determined test point: the variance of f needs the quadratic form to ~1e-6,
                # the float32 solve delivers 3e-4.  Round 5 refined the n_test solves (a second solve + a float64 product: 3.7 x the time); the
                # quadratic form is what is wanted, and X^T (2 B - K_hat X) with K_hat X in float64 has it to SECOND order in the solve error --
                # one float64 product, no second solve (bbmm.variational_inv_quad)
                from .bbmm import variational_inv_quad

                if train_train_covar.float64_product_available():
                    with settings.rhs_refinement(False), settings.cg_tolerance(settings.cg_tolerance.value() * settings.rhs_refinement.variance_tolerance_factor):
                        rhs = train_train_covar.solve(rhs_in)
                    quad = variational_inv_quad(train_train_covar.matmul_float64, rhs_in, rhs)
                    if torch.is_tensor(test_test_covar):
                        return to_linear_operator((test_test_covar.to(torch.float64) - quad).to(ttc.dtype))
                    # (K_** is an operator above max_eager_kernel_size: its own diagonal / products stay matrix-free; the n_test x n_test
                    # correction is dense, kept in float64 so that diag(K_**) - diag(quad) loses nothing before the final rounding)
                    return _VarianceDifference(test_test_covar, quad)
            rhs = train_train_covar.solve(rhs_in)
            if torch.is_tensor(test_test_covar):
                if settings.rhs_refinement.on() and ttc.dtype == torch.float32:
                    # (operators without a fused float64 product: the solves were refined inside solve(); the LAST contraction must not throw
                    # the digits away again -- float64 GEMM, float32 result)
                    return to_linear_operator((test_test_covar.to(torch.float64) - ttc.to(torch.float64) @ rhs.to(torch.float64)).to(ttc.dtype))
                return to_linear_operator(test_test_covar - ttc @ rhs)
            return test_test_covar + MatmulLinearOperator(DenseLinearOperator(ttc), DenseLinearOperator(rhs.mul(-1)))
        root = to_linear_operator(test_train_covar) @ self.covar_cache  # [n_test, m]
        if torch.is_tensor(test_test_covar):
            return to_linear_operator(torch.add(test_test_covar, root @ root.mT, alpha=-1))
        return SumLinearOperator(test_test_covar, MatmulLinearOperator(DenseLinearOperator(root), DenseLinearOperator(root.mT.mul(-1))))


class _VarianceDifference(SumLinearOperator):
    """K_** - Q with K_** a (matrix-free) operator and Q a dense float64 [m, m] matrix (``bbmm.variational_inv_quad``): behaves like the sum, but
    the diagonal is formed in float64 (the predictive variance of f is a difference of nearly equal numbers)."""

    def __init__(self, kss, quad64: torch.Tensor):
        super().__init__(kss, DenseLinearOperator((-quad64).to(kss.dtype)))
        self._kss, self._quad64 = kss, quad64

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return (self._kss.diagonal().to(torch.float64) - self._quad64.diagonal()).to(self._kss.dtype)


def prediction_strategy(train_inputs, train_prior_dist, train_labels, likelihood):
    """exact_prediction_strategies.py:30-36."""
    return DefaultPredictionStrategy(train_inputs, train_prior_dist, train_labels, likelihood)


class GP(Module):
    pass


class ExactGP(GP):
    def __init__(self, train_inputs, train_targets, likelihood):
        if train_inputs is not None and torch.is_tensor(train_inputs):
            train_inputs = (train_inputs,)
        if train_inputs is not None and not all(torch.is_tensor(t) for t in train_inputs):
            raise RuntimeError("Train inputs must be a tensor, or a list/tuple of tensors")
        super().__init__()
        if train_inputs is not None:
            self.train_inputs = tuple(t.unsqueeze(-1) if t.ndimension() == 1 else t for t in train_inputs)
            self.train_targets = train_targets
        else:
            self.train_inputs = None
            self.train_targets = None
        self.likelihood = likelihood
        self.prediction_strategy = None

    def set_train_data(self, inputs=None, targets=None, strict=True):
        """exact_gp.py:113-149."""
        if inputs is not None:
            if torch.is_tensor(inputs):
                inputs = (inputs,)
            inputs = tuple(i.unsqueeze(-1) if i.ndimension() == 1 else i for i in inputs)
            if strict:
                for new, old in zip(inputs, self.train_inputs or (None,)):
                    if old is not None and (new.shape != old.shape or new.dtype != old.dtype or new.device != old.device):
                        raise RuntimeError("Cannot modify shape/dtype/device of train inputs with strict=True")
            self.train_inputs = inputs
        if targets is not None:
            if strict and self.train_targets is not None and targets.shape != self.train_targets.shape:
                raise RuntimeError("Cannot modify shape of train targets with strict=True")
            self.train_targets = targets
        self.prediction_strategy = None

    def train(self, mode=True):
        if mode:
            self.prediction_strategy = None  # exact_gp.py:99-101
        return super().train(mode)

    def __call__(self, *args, **kwargs):
        """exact_gp.py:265-333."""
        train_inputs = list(self.train_inputs) if self.train_inputs is not None else []
        inputs = [i.unsqueeze(-1) if i.ndimension() == 1 else i for i in args]
        if self.training:
            if self.train_inputs is None:
                raise RuntimeError("train_inputs, train_targets cannot be None in training mode. Call .eval() for prior predictions, or call .set_train_data() to add training data.")
            if settings.debug.on():
                if not all(torch.equal(ti, inp) for ti, inp in zip(train_inputs, inputs)):
                    raise RuntimeError("You must train on the training inputs!")
            return Module.__call__(self, *inputs, **kwargs)
        if settings.prior_mode.on() or self.train_inputs is None or self.train_targets is None:
            return Module.__call__(self, *inputs, **kwargs)  # prior mode (exact_gp.py:285)
        if settings.debug.on():
            if all(torch.equal(ti, inp) for ti, inp in zip(train_inputs, inputs)):
                warnings.warn("The input matches the stored training data. Did you forget to call model.train()?", GPInputWarning)
        if self.prediction_strategy is None:
            train_output = Module.__call__(self, *train_inputs, **kwargs)
            self.prediction_strategy = prediction_strategy(train_inputs, train_output, self.train_targets, self.likelihood)
        full_mean, test_test_covar, test_train_covar = self._get_test_prior_mean_and_covariances(train_inputs, inputs, **kwargs)
        with settings.cg_tolerance(settings.eval_cg_tolerance.value()):  # exact_gp.py:324
            predictive_mean, predictive_covar = self.prediction_strategy.exact_prediction(full_mean, test_test_covar, test_train_covar)
        # exact_gp.py:330-333: reshape to the (possibly multitask n x T) event shape of the prior
        cls, tail = self._posterior_class
        if tail:
            predictive_mean = predictive_mean.view(-1, *tail)
        return cls(predictive_mean, predictive_covar)

    def get_fantasy_model(self, inputs, targets, **kwargs):
        """``exact_gp.py:151-263`` / ``exact_prediction_strategies.py:137-265`` (single-output, non-batch): a new model
        conditioned on the training data PLUS (inputs, targets), without re-solving the n x n system.  With
        B = K_hat^-1 k (one mBCG solve with m right-hand sides -- the MFMA path), S = k_hat_new - k^T B (m x m, dense
        Cholesky), alpha = the current mean cache and e = (targets - mean_new) - k^T alpha, the new mean cache is
        [alpha - B S^-1 e ; S^-1 e].  An existing LOVE covariance cache (root R of K_hat^-1) is UPDATED, not rebuilt: the
        bordered inverse gives the root [[R, -B L_s^-T], [0, L_s^-T]] (the reference: ``cat_rows`` + ``root_inv_decomposition``).
        Hyper-parameters are shared with this model, as in the reference."""
        import copy

        if self.prediction_strategy is None:
            raise RuntimeError("Fantasy observations can only be added after making predictions with a model so that "
                               "all test independent caches exist. Call the model on some data first!")
        if torch.is_tensor(inputs):
            inputs = (inputs,)
        inputs = [i.unsqueeze(-1) if i.ndimension() == 1 else i for i in inputs]
        if targets.dim() != 1 or len(self.prediction_strategy._train_shape) != 1:
            raise NotImplementedError("get_fantasy_model: single-output, non-batch models")
        train_inputs = list(self.train_inputs)
        ps = self.prediction_strategy
        n, m = ps.num_train, targets.shape[-1]
        full_inputs = [torch.cat([ti, inp], dim=-2) for ti, inp in zip(train_inputs, inputs)]
        full_targets = torch.cat([self.train_targets, targets], dim=-1)
        fantasy_kwargs = {"noise": kwargs.pop("noise")} if "noise" in kwargs else {}      # exact_gp.py:229-232: the fantasy points' own noise
        with torch.no_grad():
            full_output = Module.__call__(self, *full_inputs, **kwargs)
            prior_covar = full_output.lazy_covariance_matrix
            k = to_dense(prior_covar[:n, n:].evaluate_kernel())                               # [n, m] prior cross-covariance
            new_prior = full_output.__class__(full_output.loc[..., n:], to_dense(prior_covar[n:, n:].evaluate_kernel()))
            khat_new = to_dense(self.likelihood(new_prior, inputs, **fantasy_kwargs).lazy_covariance_matrix)   # [m, m] incl. noise
            with settings.cg_tolerance(settings.eval_cg_tolerance.value()):
                alpha = ps.mean_cache                                                         # [n]
                Bm = ps.lik_train_train_covar.solve(k)                                        # [n, m]
            S = (khat_new - k.mT @ Bm).to(torch.float64)
            e = ((targets - full_output.loc[..., n:]) - k.mT @ alpha).to(torch.float64)
            Ls = torch.linalg.cholesky(0.5 * (S + S.mT))
            w = torch.cholesky_solve(e.unsqueeze(-1), Ls).squeeze(-1).to(alpha.dtype)      # S^-1 e
            new_cache = torch.cat([alpha - Bm @ w, w], dim=-1)
            # LOVE cache (exact_prediction_strategies.py:233-238: cat_rows + root_inv_decomposition): with R R^T ~= K_hat^-1 and the
            # bordered inverse  [K_hat k; k^T k_new]^-1 = [R;0][R;0]^T + [-B; I] S^-1 [-B; I]^T,  S = L_s L_s^T,  a root of the new
            # inverse is  [[R, -B L_s^-T], [0, L_s^-T]]  -- m extra columns, no new Lanczos run
            new_covar_cache = None
            if ps._covar_cache is not None:
                R = ps._covar_cache                                                          # [n, r]
                lsinv_t = torch.linalg.solve_triangular(Ls, torch.eye(m, device=Ls.device, dtype=Ls.dtype), upper=False).mT.to(R.dtype)
                top = torch.cat([R, -(Bm.to(R.dtype) @ lsinv_t)], dim=-1)
                bot = torch.cat([torch.zeros(m, R.shape[-1], device=R.device, dtype=R.dtype), lsinv_t], dim=-1)
                new_covar_cache = torch.cat([top, bot], dim=-2)
        new_model = copy.copy(self)                 # shares parameters / modules (exact_gp.py:244-263 deep-copies only the caches)
        new_model._modules = dict(self._modules)    # (its own module table: the fantasy likelihood below must not replace this model's)
        new_model.likelihood = self.likelihood.get_fantasy_likelihood(**fantasy_kwargs)     # exact_gp.py:251
        new_model.train_inputs = tuple(full_inputs)
        new_model.train_targets = full_targets
        new_model.prediction_strategy = prediction_strategy(full_inputs, full_output, full_targets, new_model.likelihood)
        new_model.prediction_strategy._mean_cache = new_cache.detach()
        if new_covar_cache is not None:
            new_model.prediction_strategy._covar_cache = new_covar_cache.detach()
        return new_model

    def _get_test_prior_mean_and_covariances(self, train_inputs, inputs, **kwargs):
        """exact_gp.py:355-430: joint train u test prior, sliced lazily.  Batch dimensions of the training and the test
        inputs are broadcast against each other first (exact_gp.py:303-313)."""
        full_inputs = []
        for ti, inp in zip(train_inputs, inputs):
            batch = torch.broadcast_shapes(ti.shape[:-2], inp.shape[:-2])
            full_inputs.append(torch.cat([ti.expand(*batch, *ti.shape[-2:]), inp.expand(*batch, *inp.shape[-2:])], dim=-2))
        full_output = Module.__call__(self, *full_inputs, **kwargs)
        full_mean, full_covar = full_output.loc, full_output.lazy_covariance_matrix
        self._posterior_class = (full_output.__class__, tuple(full_output.event_shape[1:]))
        n = self.prediction_strategy.num_train
        test_mean = full_mean[..., n:]
        test_test_covar = full_covar[..., n:, n:].evaluate_kernel()
        test_train_covar = full_covar[..., n:, :n].evaluate_kernel()
        return test_mean, test_test_covar, test_train_covar


_ = LinearOperator
