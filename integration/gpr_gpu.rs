//! `EstimatorGpu` / `SurrogateModelGpu`: hbetune's GP surrogate on an MI355X through libhbegp.so.
//!
//! Drop-in for `EstimatorGPR` / `SurrogateModelGPR` (src/core/gpr.rs:54-63, 215-338): implements the same two traits,
//! `Estimator<A>` and `SurrogateModel<A>` (src/core/surrogate_model.rs:6-65), so `Minimizer::minimize` and everything
//! above it stay untouched.  Everything below `FittedKernel::new/extend` (src/gpr/fit.rs:18-68) and `predict`
//! (src/gpr/predict.rs:7-52) runs in hand-written HIP kernels; what stays here is the O(n) host work the reference
//! keeps in its adapter: y-normalisation, the amplitude heuristic, defaults and prior reuse, expected improvement,
//! summary statistics.
//!
//! STATUS: written out in full, NEVER COMPILED -- the repository that ships it has no Rust toolchain in its build
//! image (cargo / rustc absent).  It targets the reference at the commit surveyed (ndarray 0.13, statrs 0.12,
//! edition 2018).  Behaviour is pinned by the Python mirror of the same adapter (hbetune_rs_amd/estimator.py), which is
//! tested against the reference's own suites (tests/test_gpu_estimator.py restates tests/gpr_tests.rs).
//!
//! Installation in the hbetune crate:
//!   1. copy this file to `src/core/gpr_gpu.rs`, integration/build.rs to `build.rs`;
//!   2. `src/core/mod.rs`:  `#[cfg(feature = "gpu")] pub mod gpr_gpu;`
//!      `src/lib.rs`:       `#[cfg(feature = "gpu")] pub use crate::core::gpr_gpu::{EstimatorGpu, SurrogateModelGpu};`
//!   3. `src/bin/hbetune/main.rs:240-244`: select `command_run::<A, EstimatorGpu>` behind a `--gpu` flag
//!      (the function is already generic over the estimator: minimize.rs:231-241).
//! The one addition to the trait surface that pays on a GPU is `predict_confidence_bound_a` (batched confidence bound for
//! `find_best_individual_by_confidence_bound`, minimize.rs:680-714); it is provided as an inherent method.

use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_float, c_int};

use ndarray::prelude::*;

use crate::core::acquisition::expected_improvement;
use crate::core::surrogate_model::SummaryStatistics;
use crate::core::ynormalize::{Projection, YNormalize};
use crate::util::{BoundedValue, BoundsError};
use crate::{Estimator, Scalar, Space, SurrogateModel, RNG};

// ------------------------------------------------------------------------------------------------------------------
// FFI: 1:1 with include/hbegp.h (ABI 0.2.0)
// ------------------------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct HbegpCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct HbegpModel {
    _private: [u8; 0],
}

/// `hbegp_fit_options` (include/hbegp.h)
#[repr(C)]
pub struct HbegpFitOptions {
    /// `size_of::<HbegpFitOptions>()`: the library copies min(struct_size, its own size), so the struct may grow at its end
    pub struct_size: usize,
    pub maxeval: c_int,
    pub fixed_work: c_int,
    pub lbfgs_memory: c_int,
    pub trace_cap: c_int,
    pub trace_theta: *mut c_double,
    pub trace_lml: *mut c_double,
    pub trace_grad: *mut c_double,
    pub trace_run: *mut c_int,
    pub trace_count: *mut c_int,
    pub n_evals: *mut c_int,
    pub n_not_pd: *mut c_int,
}

pub const HBEGP_OK: c_int = 0;
pub const HBEGP_NOT_PD: c_int = 1;
pub const HBEGP_ALL_FAILED: c_int = 2;

extern "C" {
    fn hbegp_version() -> c_int;
    fn hbegp_device_count() -> c_int;
    fn hbegp_ctx_create(n_devices: c_int, device_ids: *const c_int, out: *mut *mut HbegpCtx) -> c_int;
    fn hbegp_ctx_destroy(ctx: *mut HbegpCtx);
    fn hbegp_last_error() -> *const c_char;

    fn hbegp_fit_f64(
        ctx: *mut HbegpCtx, x: *const c_double, y: *const c_double, n: c_int, d: c_int, nu: c_double,
        theta0: *const c_double, lo: *const c_double, hi: *const c_double, starts: *const c_double, n_restarts: c_int,
        opt: *const HbegpFitOptions, theta_best: *mut c_double, lml_best: *mut c_double, model: *mut *mut HbegpModel,
    ) -> c_int;
    fn hbegp_fit_f32(
        ctx: *mut HbegpCtx, x: *const c_float, y: *const c_float, n: c_int, d: c_int, nu: c_double,
        theta0: *const c_double, lo: *const c_double, hi: *const c_double, starts: *const c_double, n_restarts: c_int,
        opt: *const HbegpFitOptions, theta_best: *mut c_double, lml_best: *mut c_double, model: *mut *mut HbegpModel,
    ) -> c_int;
    fn hbegp_extend_f64(
        ctx: *mut HbegpCtx, x: *const c_double, y: *const c_double, n: c_int, d: c_int, nu: c_double,
        theta: *const c_double, lo: *const c_double, hi: *const c_double, model: *mut *mut HbegpModel,
    ) -> c_int;
    fn hbegp_extend_f32(
        ctx: *mut HbegpCtx, x: *const c_float, y: *const c_float, n: c_int, d: c_int, nu: c_double,
        theta: *const c_double, lo: *const c_double, hi: *const c_double, model: *mut *mut HbegpModel,
    ) -> c_int;
    fn hbegp_extend_from_f64(
        ctx: *mut HbegpCtx, prior: *mut HbegpModel, x: *const c_double, y: *const c_double, n: c_int,
        model: *mut *mut HbegpModel, incremental: *mut c_int,
    ) -> c_int;
    fn hbegp_extend_from_f32(
        ctx: *mut HbegpCtx, prior: *mut HbegpModel, x: *const c_float, y: *const c_float, n: c_int,
        model: *mut *mut HbegpModel, incremental: *mut c_int,
    ) -> c_int;
    fn hbegp_predict_f64(
        model: *mut HbegpModel, xs: *const c_double, m: c_int, mean: *mut c_double, var: *mut c_double, n_warn: *mut c_int,
    ) -> c_int;
    fn hbegp_predict_f32(
        model: *mut HbegpModel, xs: *const c_float, m: c_int, mean: *mut c_float, var: *mut c_float, n_warn: *mut c_int,
    ) -> c_int;
    fn hbegp_model_info(
        model: *const HbegpModel, n: *mut c_int, d: *mut c_int, is_f32: *mut c_int, nu: *mut c_double, lml: *mut c_double,
    ) -> c_int;
    fn hbegp_model_get_f64(model: *mut HbegpModel, theta: *mut c_double, alpha: *mut c_double, kinv: *mut c_double) -> c_int;
    fn hbegp_model_get_f32(model: *mut HbegpModel, theta: *mut c_double, alpha: *mut c_float, kinv: *mut c_float) -> c_int;
    fn hbegp_model_retain(model: *mut HbegpModel);
    fn hbegp_model_release(model: *mut HbegpModel);
}

fn last_error() -> String {
    unsafe {
        let p = hbegp_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// A = f64 | f32 dispatch (src/gpr/scalar.rs:3-30; `--use-32`, src/bin/hbetune/main.rs:240-244)
// ------------------------------------------------------------------------------------------------------------------
mod sealed {
    pub trait Sealed {}
    impl Sealed for f64 {}
    impl Sealed for f32 {}
}

/// The two element types libhbegp carries.  Sealed: the library has exactly these two instantiations.
pub trait GpuScalar: Scalar + sealed::Sealed {
    /// `hbegp_fit_*`
    #[allow(clippy::too_many_arguments)]
    unsafe fn ffi_fit(
        ctx: *mut HbegpCtx, x: *const Self, y: *const Self, n: c_int, d: c_int, nu: f64, theta0: *const f64,
        lo: *const f64, hi: *const f64, starts: *const f64, n_restarts: c_int, opt: *const HbegpFitOptions,
        theta_best: *mut f64, lml_best: *mut f64, model: *mut *mut HbegpModel,
    ) -> c_int;
    /// `hbegp_extend_*`
    #[allow(clippy::too_many_arguments)]
    unsafe fn ffi_extend(
        ctx: *mut HbegpCtx, x: *const Self, y: *const Self, n: c_int, d: c_int, nu: f64, theta: *const f64,
        lo: *const f64, hi: *const f64, model: *mut *mut HbegpModel,
    ) -> c_int;
    /// `hbegp_extend_from_*`
    unsafe fn ffi_extend_from(
        ctx: *mut HbegpCtx, prior: *mut HbegpModel, x: *const Self, y: *const Self, n: c_int,
        model: *mut *mut HbegpModel, incremental: *mut c_int,
    ) -> c_int;
    /// `hbegp_predict_*`
    unsafe fn ffi_predict(
        model: *mut HbegpModel, xs: *const Self, m: c_int, mean: *mut Self, var: *mut Self, n_warn: *mut c_int,
    ) -> c_int;
}

impl GpuScalar for f64 {
    unsafe fn ffi_fit(
        ctx: *mut HbegpCtx, x: *const f64, y: *const f64, n: c_int, d: c_int, nu: f64, theta0: *const f64,
        lo: *const f64, hi: *const f64, starts: *const f64, n_restarts: c_int, opt: *const HbegpFitOptions,
        theta_best: *mut f64, lml_best: *mut f64, model: *mut *mut HbegpModel,
    ) -> c_int {
        hbegp_fit_f64(ctx, x, y, n, d, nu, theta0, lo, hi, starts, n_restarts, opt, theta_best, lml_best, model)
    }
    unsafe fn ffi_extend(
        ctx: *mut HbegpCtx, x: *const f64, y: *const f64, n: c_int, d: c_int, nu: f64, theta: *const f64,
        lo: *const f64, hi: *const f64, model: *mut *mut HbegpModel,
    ) -> c_int {
        hbegp_extend_f64(ctx, x, y, n, d, nu, theta, lo, hi, model)
    }
    unsafe fn ffi_extend_from(
        ctx: *mut HbegpCtx, prior: *mut HbegpModel, x: *const f64, y: *const f64, n: c_int,
        model: *mut *mut HbegpModel, incremental: *mut c_int,
    ) -> c_int {
        hbegp_extend_from_f64(ctx, prior, x, y, n, model, incremental)
    }
    unsafe fn ffi_predict(
        model: *mut HbegpModel, xs: *const f64, m: c_int, mean: *mut f64, var: *mut f64, n_warn: *mut c_int,
    ) -> c_int {
        hbegp_predict_f64(model, xs, m, mean, var, n_warn)
    }
}

impl GpuScalar for f32 {
    unsafe fn ffi_fit(
        ctx: *mut HbegpCtx, x: *const f32, y: *const f32, n: c_int, d: c_int, nu: f64, theta0: *const f64,
        lo: *const f64, hi: *const f64, starts: *const f64, n_restarts: c_int, opt: *const HbegpFitOptions,
        theta_best: *mut f64, lml_best: *mut f64, model: *mut *mut HbegpModel,
    ) -> c_int {
        hbegp_fit_f32(ctx, x, y, n, d, nu, theta0, lo, hi, starts, n_restarts, opt, theta_best, lml_best, model)
    }
    unsafe fn ffi_extend(
        ctx: *mut HbegpCtx, x: *const f32, y: *const f32, n: c_int, d: c_int, nu: f64, theta: *const f64,
        lo: *const f64, hi: *const f64, model: *mut *mut HbegpModel,
    ) -> c_int {
        hbegp_extend_f32(ctx, x, y, n, d, nu, theta, lo, hi, model)
    }
    unsafe fn ffi_extend_from(
        ctx: *mut HbegpCtx, prior: *mut HbegpModel, x: *const f32, y: *const f32, n: c_int,
        model: *mut *mut HbegpModel, incremental: *mut c_int,
    ) -> c_int {
        hbegp_extend_from_f32(ctx, prior, x, y, n, model, incremental)
    }
    unsafe fn ffi_predict(
        model: *mut HbegpModel, xs: *const f32, m: c_int, mean: *mut f32, var: *mut f32, n_warn: *mut c_int,
    ) -> c_int {
        hbegp_predict_f32(model, xs, m, mean, var, n_warn)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Context: devices, shared by every model of a run.  Reference-counted so that estimator and models can be cloned/moved.
// ------------------------------------------------------------------------------------------------------------------
struct Context {
    raw: *mut HbegpCtx,
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { hbegp_ctx_destroy(self.raw) }
    }
}
impl Context {
    /// All visible gfx950 devices: the optimiser runs of a fit are sharded run r -> device r mod G (gradmin.rs:19-31 is
    /// the axis; no collective).  `HBEGP_DEVICES=k` limits the count.
    fn open() -> Result<std::rc::Rc<Context>, Error> {
        let abi = unsafe { hbegp_version() };
        if abi < 200 {
            return Err(Error::Backend(format!("libhbegp ABI {} is older than this binding (200: hbegp_fit_options.struct_size)", abi)));
        }
        let mut count = unsafe { hbegp_device_count() };
        if let Some(limit) = std::env::var("HBEGP_DEVICES").ok().and_then(|s| s.parse::<c_int>().ok()) {
            count = count.min(limit.max(1));
        }
        if count < 1 {
            return Err(Error::Backend("no gfx950 device visible: libhbegp has no CPU fallback".to_owned()));
        }
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { hbegp_ctx_create(count, std::ptr::null(), &mut raw) };
        if rc != HBEGP_OK {
            return Err(Error::Backend(last_error()));
        }
        Ok(std::rc::Rc::new(Context { raw }))
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Model
// ------------------------------------------------------------------------------------------------------------------
/// Mirror of `SurrogateModelGPR<A>` (gpr.rs:54-63).  X, alpha, K^-1 live on the device behind `handle`; the host keeps
/// what `get_kernel_or_default` hands to the next fit (kernel parameters AND their bounds, gpr.rs:407-409).
pub struct SurrogateModelGpu<A: Scalar> {
    handle: *mut HbegpModel,
    ctx: std::rc::Rc<Context>,
    /// `[ln s2, ln c, ln ell_1..]`, clamped into the bounds (fit.rs:155-164)
    theta: Vec<f64>,
    /// linear-space bounds in theta order: noise, amplitude, length scales
    lo: Vec<f64>,
    hi: Vec<f64>,
    matern_nu: f64,
    y_norm: YNormalize<A>,
    lml: f64,
}

impl<A: Scalar> std::fmt::Debug for SurrogateModelGpu<A> {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        f.debug_struct("SurrogateModelGpu")
            .field("noise", &self.theta[0].exp())
            .field("amplitude", &self.theta[1].exp())
            .field("length_scale", &self.theta[2..].iter().map(|t| t.exp()).collect::<Vec<_>>())
            .field("matern_nu", &self.matern_nu)
            .field("lml", &self.lml)
            .field("y_norm", &self.y_norm)
            .finish()
    }
}

/// Models are kept for the whole run (`all_models`, minimize.rs:331) and `SurrogateModelGPR` is `Clone` (gpr.rs:53):
/// the device copy is shared and reference-counted by the library, freed on the last drop.
impl<A: Scalar> Clone for SurrogateModelGpu<A> {
    fn clone(&self) -> Self {
        unsafe { hbegp_model_retain(self.handle) };
        SurrogateModelGpu {
            handle: self.handle,
            ctx: self.ctx.clone(),
            theta: self.theta.clone(),
            lo: self.lo.clone(),
            hi: self.hi.clone(),
            matern_nu: self.matern_nu,
            y_norm: self.y_norm.clone(),
            lml: self.lml,
        }
    }
}

impl<A: Scalar> Drop for SurrogateModelGpu<A> {
    fn drop(&mut self) {
        unsafe { hbegp_model_release(self.handle) }
    }
}

impl<A: GpuScalar> SurrogateModelGpu<A> {
    /// log marginal likelihood of the captured evaluation (`FittedKernel::lml`, fit.rs:11)
    pub fn lml(&self) -> f64 {
        self.lml
    }

    pub fn noise(&self) -> BoundedValue<f64> {
        BoundedValue::new(self.theta[0].exp(), self.lo[0], self.hi[0]).expect("fitted noise lies in its bounds")
    }

    pub fn amplitude(&self) -> BoundedValue<f64> {
        BoundedValue::new(self.theta[1].exp(), self.lo[1], self.hi[1]).expect("fitted amplitude lies in its bounds")
    }

    /// alpha (n) and the full symmetric K^-1 (n x n) copied back from the device -- `FittedKernel::{alpha, k_inv}`
    /// (fit.rs:6-12); only needed by code that wants the CPU `predict` beside the device one.
    pub fn arrays(&self) -> (Array1<A>, Array2<A>)
    where
        A: HostCopy,
    {
        let (mut n, mut d) = (0 as c_int, 0 as c_int);
        let rc = unsafe {
            hbegp_model_info(self.handle, &mut n, &mut d, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())
        };
        assert!(rc == HBEGP_OK, "hbegp_model_info: {}", last_error());
        let n = n as usize;
        let mut alpha = Array1::<A>::zeros(n);
        let mut kinv = Array2::<A>::zeros((n, n));
        let rc = unsafe { A::ffi_model_get(self.handle, alpha.as_mut_ptr(), kinv.as_mut_ptr()) };
        assert!(rc == HBEGP_OK, "hbegp_model_get: {}", last_error());
        (alpha, kinv)
    }

    /// `predict()` (predict.rs:7-52) on the device: normalised mean and, if wanted, variance
    /// (`c + 1e-5 - k*^T K^-1 k*`, negatives clamped to 0).  Prints the reference's warning when variances fell below
    /// `-sqrt(1e-5)` before clamping (predict.rs:39-48).
    fn predict_normalized(&self, x: ArrayView2<A>, want_variance: bool) -> (Array1<A>, Option<Array1<A>>) {
        let m = x.nrows();
        let x = x.as_standard_layout(); // row-major m x d, as `project_into_features_array` produces (space.rs:141-159)
        let mut mean = Array1::<A>::zeros(m);
        let mut var = if want_variance { Some(Array1::<A>::zeros(m)) } else { None };
        if m == 0 {
            return (mean, var);
        }
        let mut n_warn: c_int = 0;
        let var_ptr = var.as_mut().map(|v| v.as_mut_ptr()).unwrap_or(std::ptr::null_mut());
        let rc = unsafe { A::ffi_predict(self.handle, x.as_ptr(), m as c_int, mean.as_mut_ptr(), var_ptr, &mut n_warn) };
        if rc != HBEGP_OK {
            panic!("hbegp_predict failed: {}", last_error());
        }
        if n_warn > 0 {
            eprintln!("Variances below 0 were predicted and will be corrected ({} values)", n_warn);
        }
        (mean, var)
    }

    /// Batched `predict_confidence_bound` (gpr.rs:94-112 for every row): one device call instead of one per individual
    /// in `find_best_individual_by_confidence_bound` (minimize.rs:680-714).
    pub fn predict_confidence_bound_a(&self, x: Array2<A>, cb: A) -> Array1<A> {
        let (mnorm, vnorm) = self.predict_normalized(x.view(), true);
        let stdnorm = vnorm.expect("variance was requested").mapv(|v| v.sqrt());
        // Confidence bounds are quantiles of the distribution, thus treating them as fixed locations is correct.
        self.y_norm.project_location_from_normalized(mnorm + stdnorm * cb)
    }
}

/// `hbegp_model_get_*` per element type (kept apart from `GpuScalar` so that the hot trait stays minimal).
pub trait HostCopy: GpuScalar {
    unsafe fn ffi_model_get(model: *mut HbegpModel, alpha: *mut Self, kinv: *mut Self) -> c_int;
}
impl HostCopy for f64 {
    unsafe fn ffi_model_get(model: *mut HbegpModel, alpha: *mut f64, kinv: *mut f64) -> c_int {
        hbegp_model_get_f64(model, std::ptr::null_mut(), alpha, kinv)
    }
}
impl HostCopy for f32 {
    unsafe fn ffi_model_get(model: *mut HbegpModel, alpha: *mut f32, kinv: *mut f32) -> c_int {
        hbegp_model_get_f32(model, std::ptr::null_mut(), alpha, kinv)
    }
}

impl<A: GpuScalar> SurrogateModel<A> for SurrogateModelGpu<A> {
    /// gpr.rs:72-79
    fn length_scales(&self) -> Vec<f64> {
        self.theta[2..].iter().map(|t| t.exp()).collect()
    }

    /// gpr.rs:81-92
    fn predict_mean_a(&self, x: Array2<A>) -> Array1<A> {
        let (y, _) = self.predict_normalized(x.view(), false);
        self.y_norm.project_location_from_normalized(y)
    }

    /// gpr.rs:94-112
    fn predict_confidence_bound(&self, x: Array1<A>, cb: A) -> A {
        let ys = self.predict_confidence_bound_a(x.insert_axis(Axis(0)), cb);
        *ys.first().unwrap()
    }

    /// Summary of the predictive distribution at one point -- the behaviour of gpr.rs:114-177, written independently of its text:
    /// mean / std / cv through `YNormalize`'s projections; the quartiles are the normal quantiles of N(m, s) in normalised units
    /// (all three equal m when s vanishes), projected back as locations.  Panics where the reference panics.
    fn predict_statistics(&self, x: Array1<A>) -> SummaryStatistics<A> {
        let (m, v) = self.predict_normalized(x.view().insert_axis(Axis(0)), true);
        let v = v.expect("variance was requested");
        assert!(m.len() == 1 && v.len() == 1, "should contain one element");
        let m0: f64 = m[0].into();
        let s0: f64 = v[0].sqrt().into();
        let quantiles_norm: [f64; 3] = if abs_diff_eq!(s0, 0.0) {
            [m0; 3]
        } else {
            use statrs::distribution::InverseCDF;
            let dist = statrs::distribution::Normal::new(m0, s0).unwrap_or_else(|err| {
                panic!("could not create normal distribution with mean {} std {}: {}", m0, s0, err)
            });
            [dist.inverse_cdf(0.25), dist.inverse_cdf(0.5), dist.inverse_cdf(0.75)]
        };
        let yn = &self.y_norm;
        let mean = yn.project_mean_from_normalized(m.clone(), v.view())[0];
        let std = yn.project_std_from_normalized(m.view(), v.clone())[0];
        let cv = yn.project_cv_from_normalized(m.view(), v)[0];
        let q = yn.project_location_from_normalized(quantiles_norm.iter().map(|&q| A::from_f(q)).collect::<Array1<A>>());
        SummaryStatistics::new_mean_std_cv_quartiles(mean, std, cv, [q[0], q[1], q[2]])
    }

    /// Mean and expected improvement for a batch (behaviour of gpr.rs:179-212) -- the entry the acquisition should use, one call
    /// per generation (hbetune_rs_amd/estimator.py::acquire_by_mutation re-expresses acquisition.rs:86-116, 177-202 over it):
    /// the incumbent goes into normalised units, EI is taken there per candidate on (mean, std), the mean comes back as a location.
    fn predict_mean_ei_a(&self, x: Array2<A>, fmin: A) -> (Array1<A>, Array1<A>) {
        let (mean_n, var_n) = self.predict_normalized(x.view(), true);
        let var_n = var_n.expect("variance was requested");
        let fmin_n: f64 = self.y_norm.project_into_normalized(array![fmin])[0].into();
        let ei: Array1<A> = mean_n
            .iter()
            .zip(var_n.iter())
            .map(|(&m, &v)| A::from_f(expected_improvement(m.into(), v.sqrt().into(), fmin_n)))
            .collect();
        (self.y_norm.project_location_from_normalized(mean_n), ei)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Estimator
// ------------------------------------------------------------------------------------------------------------------
/// Mirror of `EstimatorGPR` (gpr.rs:339-400) + the device context.
#[derive(Clone)]
pub struct EstimatorGpu {
    noise_bounds: (f64, f64),
    length_scale_bounds: Vec<(f64, f64)>,
    n_restarts_optimizer: usize,
    matern_nu: f64,
    amplitude_bounds: Option<(f64, f64)>,
    y_projection: Projection,
    known_optimum: Option<f64>,
    /// evaluations per optimiser run: the reference's NLopt cap (gradmin.rs:54)
    maxeval: usize,
    ctx: std::rc::Rc<Context>,
}

impl std::fmt::Debug for EstimatorGpu {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        f.debug_struct("EstimatorGpu")
            .field("noise_bounds", &self.noise_bounds)
            .field("length_scale_bounds", &self.length_scale_bounds)
            .field("n_restarts_optimizer", &self.n_restarts_optimizer)
            .field("matern_nu", &self.matern_nu)
            .field("amplitude_bounds", &self.amplitude_bounds)
            .field("y_projection", &self.y_projection)
            .field("known_optimum", &self.known_optimum)
            .finish()
    }
}

/// builders: gpr.rs:351-400
impl EstimatorGpu {
    pub fn noise_bounds(self, lo: f64, hi: f64) -> Self {
        EstimatorGpu { noise_bounds: (lo, hi), ..self }
    }

    pub fn length_scale_bounds(self, bounds: Vec<(f64, f64)>) -> Self {
        EstimatorGpu { length_scale_bounds: bounds, ..self }
    }

    pub fn n_restarts_optimizer(self, n: usize) -> Self {
        EstimatorGpu { n_restarts_optimizer: n, ..self }
    }

    /// 0.5, 1.5 or 2.5 (matern_kernel.rs:65-80; other values are `unimplemented!` there and rejected by the library);
    /// `f64::INFINITY` selects the library's squared-exponential kernel (an extension: no CPU counterpart in hbetune)
    pub fn matern_nu(self, nu: f64) -> Self {
        EstimatorGpu { matern_nu: nu, ..self }
    }

    pub fn amplitude_bounds(self, bounds: Option<(f64, f64)>) -> Self {
        EstimatorGpu { amplitude_bounds: bounds, ..self }
    }

    pub fn y_projection(self, y_projection: Projection) -> Self {
        EstimatorGpu { y_projection, ..self }
    }

    pub fn known_optimum(self, known_optimum: f64) -> Self {
        EstimatorGpu { known_optimum: Some(known_optimum), ..self }
    }
}

/// gpr.rs:429-450, verbatim semantics (the function is private to gpr.rs, hence restated)
fn estimate_amplitude<A: Scalar>(y: ArrayView1<A>, bounds: Option<(f64, f64)>) -> BoundedValue<f64> {
    use ndarray_stats::Quantile1dExt as _;
    use noisy_float::types::N64;
    let (lo, hi) = bounds.unwrap_or_else(|| {
        let hi = y.mapv(|x| x.powi(2)).sum().into();
        let lo = y
            .mapv(|x| N64::from_f64(x.into()))
            .quantile_mut(N64::from_f64(0.1), &ndarray_stats::interpolate::Lower)
            .unwrap()
            .raw()
            .powi(2)
            * y.len() as f64;
        assert!(lo >= 0.0);
        let lo = if lo > 2e-5 { lo } else { 2e-5 };
        (lo / 2.0, hi * 2.0)
    });
    let start = f64::exp((lo.ln() + hi.ln()) / 2.0);
    BoundedValue::new(start, lo, hi).unwrap()
}

/// `get_kernel_or_default` (gpr.rs:402-427) in the library's terms: start point `[ln s2, ln c, ln ell..]` and
/// linear-space bounds in the same order (fit.rs:140-144).  A prior hands over its whole kernel: parameters, bounds, nu.
struct KernelSpec {
    theta0: Vec<f64>,
    lo: Vec<f64>,
    hi: Vec<f64>,
    nu: f64,
}

fn get_kernel_or_default<A: Scalar>(
    prior: Option<&SurrogateModelGpu<A>>,
    amplitude: BoundedValue<f64>,
    config: &EstimatorGpu,
) -> Result<KernelSpec, Error> {
    if let Some(prior) = prior {
        return Ok(KernelSpec {
            theta0: prior.theta.clone(),
            lo: prior.lo.clone(),
            hi: prior.hi.clone(),
            nu: prior.matern_nu,
        });
    }

    let noise = BoundedValue::new(1.0, config.noise_bounds.0, config.noise_bounds.1).map_err(Error::NoiseBounds)?;

    let length_scale: Vec<BoundedValue<f64>> = config
        .length_scale_bounds
        .iter()
        .map(|&(lo, hi)| BoundedValue::new(((lo.ln() + hi.ln()) / 2.).exp(), lo, hi))
        .collect::<Result<_, _>>()
        .map_err(Error::LengthScaleBounds)?;

    let mut theta0 = vec![noise.value().ln(), amplitude.value().ln()];
    let mut lo = vec![noise.min(), amplitude.min()];
    let mut hi = vec![noise.max(), amplitude.max()];
    for ls in &length_scale {
        theta0.push(ls.value().ln());
        lo.push(ls.min());
        hi.push(ls.max());
    }
    Ok(KernelSpec { theta0, lo, hi, nu: config.matern_nu })
}

impl<A: GpuScalar> Estimator<A> for EstimatorGpu {
    type Model = SurrogateModelGpu<A>;
    type Error = Error;

    /// gpr.rs:219-236 + the device context.  Panics without a usable GPU: `Estimator::new` cannot return an error
    /// (surrogate_model.rs:11) and there is no CPU fallback in the library -- use `EstimatorGPR` for that.
    fn new(space: &Space) -> Self {
        let ctx = match Context::open() {
            Ok(ctx) => ctx,
            Err(err) => panic!("EstimatorGpu: {}", err),
        };
        EstimatorGpu {
            noise_bounds: (1e-5, 1e5),
            length_scale_bounds: std::iter::repeat((1e-3, 1e3)).take(space.len()).collect(),
            n_restarts_optimizer: 2,
            matern_nu: 5. / 2.,
            amplitude_bounds: None,
            y_projection: Projection::Linear,
            known_optimum: None,
            maxeval: 150,
            ctx,
        }
    }

    /// gpr.rs:238-291
    fn estimate(
        &self,
        x: Array2<A>,
        y: Array1<A>,
        prior: Option<&Self::Model>,
        rng: &mut RNG,
    ) -> Result<Self::Model, Self::Error> {
        let (n_observations, n_features) = x.dim();
        assert!(
            y.len() == n_observations,
            "expected y values for {} observations: {}",
            n_observations,
            y,
        );

        let (y_train, y_norm) =
            YNormalize::new_project_into_normalized(y, self.y_projection, self.known_optimum.map(A::from_f));

        let amplitude = estimate_amplitude(y_train.view(), self.amplitude_bounds);
        let spec = get_kernel_or_default(prior, amplitude, self)?;
        let p = spec.theta0.len();
        assert!(p == n_features + 2, "kernel has {} parameters for {} features", p, n_features);

        // Start points of the restarts: the SAME draws the reference makes, in the same order -- one uniform value per
        // parameter in its log-bounds (gradmin.rs:21-24), from the forked RNG (gpr.rs:276).  They cross the ABI explicitly,
        // so the caller's random stream is preserved.
        let mut fit_rng = rng.fork_random_state();
        let mut starts: Vec<f64> = Vec::with_capacity(self.n_restarts_optimizer * p);
        for _ in 0..self.n_restarts_optimizer {
            for (lo, hi) in spec.lo.iter().zip(&spec.hi) {
                starts.push(fit_rng.uniform(lo.ln()..=hi.ln()));
            }
        }

        let x_train = x.as_standard_layout();
        let y_train = y_train.as_standard_layout();
        let opt = HbegpFitOptions {
            struct_size: std::mem::size_of::<HbegpFitOptions>(),
            maxeval: self.maxeval as c_int,
            fixed_work: 0,
            lbfgs_memory: 0,
            trace_cap: 0,
            trace_theta: std::ptr::null_mut(),
            trace_lml: std::ptr::null_mut(),
            trace_grad: std::ptr::null_mut(),
            trace_run: std::ptr::null_mut(),
            trace_count: std::ptr::null_mut(),
            n_evals: std::ptr::null_mut(),
            n_not_pd: std::ptr::null_mut(),
        };
        let mut theta = vec![0.0f64; p];
        let mut lml = 0.0f64;
        let mut handle: *mut HbegpModel = std::ptr::null_mut();
        let rc = unsafe {
            A::ffi_fit(
                self.ctx.raw,
                x_train.as_ptr(),
                y_train.as_ptr(),
                n_observations as c_int,
                n_features as c_int,
                spec.nu,
                spec.theta0.as_ptr(),
                spec.lo.as_ptr(),
                spec.hi.as_ptr(),
                if starts.is_empty() { std::ptr::null() } else { starts.as_ptr() },
                self.n_restarts_optimizer as c_int,
                &opt,
                theta.as_mut_ptr(),
                &mut lml,
                &mut handle,
            )
        };
        match rc {
            HBEGP_OK => Ok(SurrogateModelGpu {
                handle,
                ctx: self.ctx.clone(),
                theta,
                lo: spec.lo,
                hi: spec.hi,
                matern_nu: spec.nu,
                y_norm,
                lml,
            }),
            // `capture.replace(None).unwrap()` panics in the reference when no evaluation succeeded (fit.rs:161)
            HBEGP_ALL_FAILED => panic!("called `Option::unwrap()` on a `None` value: every evaluation of the fit failed"),
            _ => Err(Error::Backend(last_error())),
        }
    }

    /// gpr.rs:293-337: no re-fit -- the prior's kernel (parameters, bounds, nu) on the new data.  The caller appends its
    /// validation samples to the rows the prior was built on (minimize.rs:629-644), so the engine reuses the prior's
    /// factorisation for the unchanged leading 128-row blocks (`hbegp_extend_from_*`) and falls back to the full
    /// evaluation by itself when they differ; results are the same to rounding.
    fn extend(
        &self,
        x: Array2<A>,
        y: Array1<A>,
        prior: &Self::Model,
        _rng: &mut RNG,
    ) -> Result<Self::Model, Self::Error> {
        let (n_observations, n_features) = x.dim();
        assert!(
            y.len() == n_observations,
            "expected y values for {} observations: {}",
            n_observations,
            y
        );
        assert!(
            prior.theta.len() == n_features + 2,
            "prior model has {} kernel parameters for {} features",
            prior.theta.len(),
            n_features
        );
        let (y_train, y_norm) =
            YNormalize::new_project_into_normalized(y, self.y_projection, self.known_optimum.map(A::from_f));

        let x_train = x.as_standard_layout();
        let y_train = y_train.as_standard_layout();
        let mut handle: *mut HbegpModel = std::ptr::null_mut();
        let mut incremental: c_int = 0;
        let rc = unsafe {
            A::ffi_extend_from(
                self.ctx.raw,
                prior.handle,
                x_train.as_ptr(),
                y_train.as_ptr(),
                n_observations as c_int,
                &mut handle,
                &mut incremental,
            )
        };
        let rc = if rc == HBEGP_OK || rc == HBEGP_NOT_PD {
            rc
        } else {
            // e.g. the prior lives on another device: the plain full path at the prior's parameters
            unsafe {
                A::ffi_extend(
                    self.ctx.raw,
                    x_train.as_ptr(),
                    y_train.as_ptr(),
                    n_observations as c_int,
                    n_features as c_int,
                    prior.matern_nu,
                    prior.theta.as_ptr(),
                    prior.lo.as_ptr(),
                    prior.hi.as_ptr(),
                    &mut handle,
                )
            }
        };
        match rc {
            HBEGP_OK => {
                let mut lml = 0.0f64;
                let info = unsafe {
                    hbegp_model_info(handle, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), &mut lml)
                };
                assert!(info == HBEGP_OK, "hbegp_model_info: {}", last_error());
                Ok(SurrogateModelGpu {
                    handle,
                    ctx: self.ctx.clone(),
                    theta: prior.theta.clone(),
                    lo: prior.lo.clone(),
                    hi: prior.hi.clone(),
                    matern_nu: prior.matern_nu,
                    y_norm,
                    lml,
                })
            }
            HBEGP_NOT_PD => panic!("Kernel matrix must be invertible."), // fit.rs:55
            _ => Err(Error::Backend(last_error())),
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Error: gpr.rs:453-473 + the backend
// ------------------------------------------------------------------------------------------------------------------
#[derive(Debug)]
pub enum Error {
    NoiseBounds(BoundsError<f64>),
    LengthScaleBounds(BoundsError<f64>),
    /// a HIP / library failure (text of `hbegp_last_error`)
    Backend(String),
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        match *self {
            Error::NoiseBounds(BoundsError { value, min, max }) => write!(
                f,
                "noise level {} violated bounds [{}, {}] during model fitting",
                value, min, max,
            ),
            Error::LengthScaleBounds(BoundsError { value, min, max }) => write!(
                f,
                "length scale {} violated bounds [{}, {}] during model fitting",
                value, min, max,
            ),
            Error::Backend(ref msg) => write!(f, "GPU backend: {}", msg),
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Tests (need an MI355X): the reference's suites are generic over the estimator, so they run unchanged on this type --
// tests/minimize_test.rs:370-397 takes `Model: hbetune::Estimator<A>`; tests/gpr_tests.rs needs `EstimatorGPR` replaced
// by `EstimatorGpu` in its `use` line.  The two below pin what is specific to this binding.
// ------------------------------------------------------------------------------------------------------------------
#[cfg(test)]
mod tests {
    use super::*;

    fn space_1d() -> Space {
        let mut space = Space::new();
        space.add_real_parameter("test", 0.0, 1.0);
        space
    }

    #[test]
    fn clone_and_drop_share_one_device_model() {
        let xs = array![0.1, 0.5, 0.5, 0.9].insert_axis(Axis(1));
        let ys = array![1.0, 1.8, 2.2, 3.0];
        let model = <EstimatorGpu as Estimator<f64>>::new(&space_1d())
            .estimate(xs, ys, None, &mut RNG::new_with_seed(123))
            .unwrap();
        let copy = model.clone();
        let a = model.predict_mean(array![0.3]);
        drop(model);
        let b = copy.predict_mean(array![0.3]); // the device copy outlives the first owner
        assert!(a == b);
        assert_abs_diff_eq!(a, 1.5, epsilon = 0.1); // gpr_tests.rs:105-110
    }

    #[test]
    fn batched_confidence_bound_matches_the_scalar_trait_method() {
        let xs = array![0.3, 0.5, 0.7].insert_axis(Axis(1));
        let ys = array![1.0, 2.0, 1.5];
        let model = <EstimatorGpu as Estimator<f64>>::new(&space_1d())
            .noise_bounds(1e-5, 1e0)
            .length_scale_bounds(vec![(0.1, 1.0)])
            .estimate(xs, ys, None, &mut RNG::new_with_seed(9372))
            .unwrap();
        let grid = Array::linspace(0.0, 1.0, 11).insert_axis(Axis(1));
        let batched = model.predict_confidence_bound_a(grid.clone(), 1.5);
        for (row, want) in grid.outer_iter().zip(batched.iter()) {
            assert_abs_diff_eq!(model.predict_confidence_bound(row.to_owned(), 1.5), *want, epsilon = 1e-12);
        }
    }
}
