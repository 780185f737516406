//! The strategy types of the MI355X backend.  They implement the reference's own traits
//! (`Interp1DStrategyBuilder` / `Interp1DStrategy`, src/interp1d/strategies/mod.rs:12-65, and the 2-D pair,
//! src/interp2d/strategies/mod.rs:14-73) -- with `rust/patches/ndarray-interp-0.6.0-batched-hook.patch` applied, which
//! adds TWO defaulted methods per finished-strategy trait (`interp_array_into`, default body = the reference's
//! per-query loop, and `interp_array_into_owned`, default = the former, for `interp_array`'s own buffer) -- so a caller switches by naming another strategy:
//!
//! ```ignore
//! use ndarray_interp::interp1d::Interp1DBuilder;
//! use ndarray_interp_hip::{HipCubicSpline, HipLinear};
//! let interp = Interp1DBuilder::new(data).x(x).strategy(HipCubicSpline::new()).build()?;
//! let ys = interp.interp_array(&queries)?;          // one ndi_interp1d_eval call
//! ```
//!
//! Everything that computes goes through the C ABI; there is no CPU evaluation in this crate (a missing device is
//! a panic with the library's message).  Element types other than f32 / f64 are refused at build time with
//! `BuilderError::ValueError` -- the reference's own generic `Linear` / `Bilinear` are the strategies for those.
use std::any::TypeId;
use std::cell::Cell;
use std::fmt::Debug;
use std::os::raw::c_void;
use std::ptr::{null, null_mut};

use ndarray::{Array, ArrayBase, ArrayViewMut, Data, Dimension, Ix1, RemoveAxis};
use ndarray_interp::interp1d::{Interp1D, Interp1DStrategy, Interp1DStrategyBuilder};
use ndarray_interp::interp2d::{Interp2D, Interp2DStrategy, Interp2DStrategyBuilder};
use ndarray_interp::{BuilderError, InterpolateError};
use num_traits::{Num, NumCast};

use crate::hip_ffi as ffi;

// ---- device selection (the one option the reference has no notion of) ---------------------------
thread_local! { static DEVICE: Cell<i32> = const { Cell::new(0) }; }

/// The HIP device the strategies built on this thread put their tables on (0 unless set).
pub fn current_device() -> i32 {
    DEVICE.with(|d| d.get())
}
pub fn set_current_device(ordinal: i32) {
    assert!(
        ordinal >= 0 && ordinal < unsafe { ffi::ndi_device_count() },
        "device ordinal out of range"
    );
    DEVICE.with(|d| d.set(ordinal))
}

extern "C" {
    // the per-thread default stream of the HIP runtime: rayon callers get one stream per worker thread
    // (benches/bench_interp1d.rs:49-79 drives one interpolator from many threads)
    static hipStreamPerThread: *mut c_void;
}
pub(crate) fn per_thread_stream() -> *mut c_void {
    unsafe { hipStreamPerThread }
}

pub(crate) fn dtype_of<T: 'static>() -> Option<i32> {
    // the TypeId dispatch the reference itself uses (src/interp1d/mod.rs:283)
    if TypeId::of::<T>() == TypeId::of::<f64>() {
        Some(ffi::NDI_F64)
    } else if TypeId::of::<T>() == TypeId::of::<f32>() {
        Some(ffi::NDI_F32)
    } else {
        None
    }
}

/// `x = {x:#?} is not in range` with the value in the element type (linear.rs:81-83, cubic_spline.rs:799-801).
fn oob_message<T: NumCast + Debug>(axis: &str, value: f64, pretty: bool) -> String {
    let v: T = num_traits::cast(value).expect("the offending query is a value of the element type");
    if pretty {
        format!("{axis} = {v:#?} is not in range")
    } else {
        format!("{axis} = {v:?} is not in range") // bilinear.rs:72-79 uses {x:?}
    }
}

/// Owns one `ndi_interp1d` handle (device copies of x, data and the spline tables).
#[derive(Debug)]
pub struct DeviceTables1D {
    pub(crate) h: *mut ffi::ndi_interp1d,
    pub(crate) lanes: usize,
}
// `ndi_interp1d_eval` on one handle is thread-safe and re-entrant (per stream x thread scratch)
unsafe impl Send for DeviceTables1D {}
unsafe impl Sync for DeviceTables1D {}
impl Drop for DeviceTables1D {
    fn drop(&mut self) {
        unsafe { ffi::ndi_interp1d_destroy(self.h) }
    }
}

#[derive(Debug)]
pub struct DeviceTables2D {
    pub(crate) h: *mut ffi::ndi_interp2d,
    pub(crate) lanes: usize,
}
unsafe impl Send for DeviceTables2D {}
unsafe impl Sync for DeviceTables2D {}
impl Drop for DeviceTables2D {
    fn drop(&mut self) {
        unsafe { ffi::ndi_interp2d_destroy(self.h) }
    }
}

/// Rows of `buffer` as (pointer, row stride in elements): rows must be contiguous; a row-strided view passes its
/// stride, anything else is evaluated into a contiguous temporary and copied back (`Bounce`).
pub(crate) enum Rows<'a, T, D: Dimension> {
    Direct { ptr: *mut T, stride: usize },
    Bounce { tmp: Array<T, D>, dst: ArrayViewMut<'a, T, D> },
}
pub(crate) fn rows_of<'a, T: Clone + num_traits::Zero, D: Dimension + RemoveAxis>(
    mut buffer: ArrayViewMut<'a, T, D>,
    lanes: usize,
) -> Rows<'a, T, D> {
    let rows = buffer.shape()[0];
    let inner_contiguous = rows == 0 || buffer.index_axis(ndarray::Axis(0), 0).is_standard_layout();
    let stride0 = if buffer.ndim() > 0 && rows > 1 { buffer.strides()[0] } else { lanes as isize };
    if inner_contiguous && stride0 >= lanes as isize {
        Rows::Direct { ptr: buffer.as_mut_ptr(), stride: stride0 as usize }
    } else {
        Rows::Bounce { tmp: Array::zeros(buffer.raw_dim()), dst: buffer }
    }
}

fn eval_error_1d<T: NumCast + Debug>(st: i32, info: &ffi::ndi_oob_info) -> Result<(), InterpolateError> {
    match st {
        ffi::NDI_OK => Ok(()),
        ffi::NDI_OUT_OF_BOUNDS => Err(InterpolateError::OutOfBounds(oob_message::<T>("x", info.value, true))),
        // the reference panics: vector_extensions.rs:83-84
        ffi::NDI_NAN_QUERY => unimplemented!("failed to convert NaN to usize"),
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),
    }
}

/// One `ndi_interp1d_eval` call over a flattened query array (host arrays in and out).
/// `flags`: `ndi_eval_flags` -- `NDI_EVAL_FRESH_OUTPUT` when the rows belong to `Interp1D::interp_array`'s own
/// `Array::zeros` buffer (the `interp_array_into_owned` hook), else `NDI_EVAL_DEFAULT`.
fn eval_1d<T, Sq, D>(
    dev: &DeviceTables1D,
    xs: &ArrayBase<Sq, Ix1>,
    buffer: ArrayViewMut<'_, T, D>,
    flags: i32,
) -> Result<(), InterpolateError>
where
    T: Num + NumCast + Copy + Debug + 'static,
    Sq: Data<Elem = T>,
    D: Dimension + RemoveAxis,
{
    let xs = xs.as_standard_layout(); // views may be strided (tests/interp1d.rs:143-148)
    assert_eq!(buffer.shape()[0], xs.len(), "one buffer row per query");
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_HOST,
        stream: per_thread_stream(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    match rows_of(buffer, dev.lanes) {
        Rows::Direct { ptr, stride } => {
            let st = unsafe {
                ffi::ndi_interp1d_eval(dev.h, xs.as_ptr().cast(), xs.len() as u64, ptr.cast(), stride as u64, &opts, &mut info)
            };
            eval_error_1d::<T>(st, &info)
        }
        Rows::Bounce { mut tmp, mut dst } => {
            let st = unsafe {
                ffi::ndi_interp1d_eval(dev.h, xs.as_ptr().cast(), xs.len() as u64, tmp.as_mut_ptr().cast(), dev.lanes as u64, &opts, &mut info)
            };
            // rows before the first failing query are written, later rows stay untouched (interp1d/mod.rs:334-342)
            let good = if st == ffi::NDI_OK { xs.len() } else { info.index as usize };
            dst.slice_axis_mut(ndarray::Axis(0), (0..good).into())
                .assign(&tmp.slice_axis(ndarray::Axis(0), (0..good).into()));
            eval_error_1d::<T>(st, &info)
        }
    }
}

fn create_1d<T, Sx2, Sd, D>(
    strategy: i32,
    extrapolate: bool,
    device: i32,
    x: &ArrayBase<Sx2, Ix1>,
    data: &ArrayBase<Sd, D>,
    periodic: bool,
    build_flags: i32,
    left: ffi::ndi_boundary,
    right: ffi::ndi_boundary,
    lanes_bc: Option<&LaneBoundaries>,
) -> Result<DeviceTables1D, BuilderError>
where
    T: 'static,
    Sx2: Data<Elem = T>,
    Sd: Data<Elem = T>,
    D: Dimension,
    T: Clone,
{
    let Some(dtype) = dtype_of::<T>() else {
        return Err(BuilderError::ValueError(
            "the MI355X strategies cover f32 and f64; use the generic Linear / Bilinear for other element types".into(),
        ));
    };
    // Interp1DBuilder::build has validated x / data already (interp1d/mod.rs:449-471) -> validate: 0
    let x = x.as_standard_layout();
    let data = data.as_standard_layout();
    let n = data.shape()[0];
    let lanes = if n == 0 { 0 } else { data.len() / n };
    let desc = ffi::ndi_interp1d_desc {
        dtype,
        strategy,
        extrapolate: extrapolate as i32,
        device,
        n: n as u64,
        lanes: lanes as u64,
        x_len: x.len() as u64,
        x: x.as_ptr().cast(),
        data: data.as_ptr().cast(),
        memspace: ffi::NDI_MEM_HOST,
        validate: 0,
        periodic: periodic as i32,
        build_flags,
        left,
        right,
        lane_left_kind: lanes_bc.map_or(null(), |l| l.left_kind.as_ptr()),
        lane_left_value: lanes_bc.map_or(null(), |l| l.left_value.as_ptr()),
        lane_right_kind: lanes_bc.map_or(null(), |l| l.right_kind.as_ptr()),
        lane_right_value: lanes_bc.map_or(null(), |l| l.right_value.as_ptr()),
    };
    let mut h = null_mut();
    match unsafe { ffi::ndi_interp1d_create(&desc, &mut h) } {
        ffi::NDI_OK => Ok(DeviceTables1D { h, lanes }),
        ffi::NDI_NOT_ENOUGH_DATA => Err(BuilderError::NotEnoughData(ffi::last_error())),
        ffi::NDI_MONOTONIC => Err(BuilderError::Monotonic(ffi::last_error())),
        ffi::NDI_SHAPE => Err(BuilderError::ShapeError(ffi::last_error())),
        ffi::NDI_VALUE => Err(BuilderError::ValueError(ffi::last_error())), // periodic y[0] != y[n-1]
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),                 // no CPU fallback: fail loudly
    }
}

// =================================================================================================
// Linear (src/interp1d/strategies/linear.rs)
// =================================================================================================
/// `Linear` on the device.  Same builder surface as the reference's (`new`, `extrapolate`) + `.device(ordinal)`.
#[derive(Debug, Default)]
pub struct HipLinear {
    extrapolate: bool,
    device: Option<i32>,
}
impl HipLinear {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn extrapolate(mut self, extrapolate: bool) -> Self {
        self.extrapolate = extrapolate;
        self
    }
    pub fn device(mut self, ordinal: i32) -> Self {
        self.device = Some(ordinal);
        self
    }
}

/// The finished strategy: the handle that owns the device tables.
#[derive(Debug)]
pub struct HipLinearStrategy {
    pub(crate) dev: DeviceTables1D,
}

impl<Sd, Sx, D> Interp1DStrategyBuilder<Sd, Sx, D> for HipLinear
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
{
    const MINIMUM_DATA_LENGHT: usize = 2; // linear.rs:52
    type FinishedStrat = HipLinearStrategy;
    fn build<Sx2>(self, x: &ArrayBase<Sx2, Ix1>, data: &ArrayBase<Sd, D>) -> Result<Self::FinishedStrat, BuilderError>
    where
        Sx2: Data<Elem = Sd::Elem>,
    {
        let zero = ffi::ndi_boundary::default();
        let dev = create_1d::<Sd::Elem, _, _, _>(
            ffi::NDI_LINEAR,
            self.extrapolate,
            self.device.unwrap_or_else(current_device),
            x,
            data,
            false,
            ffi::NDI_BUILD_DEFAULT,
            zero,
            zero,
            None,
        )?;
        Ok(HipLinearStrategy { dev })
    }
}

impl<Sd, Sx, D> Interp1DStrategy<Sd, Sx, D> for HipLinearStrategy
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
{
    /// Single point = a batch of one (launch latency >> work: INTEGRATION.md 5b says when to stay on the CPU).
    fn interp_into(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        target: ArrayViewMut<'_, Sd::Elem, D::Smaller>,
        x: Sx::Elem,
    ) -> Result<(), InterpolateError> {
        let xs = ndarray::arr1(&[x]);
        eval_1d(&self.dev, &xs, target.insert_axis(ndarray::Axis(0)), ffi::NDI_EVAL_DEFAULT)
    }

    fn interp_array_into<Sq>(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        xs: &ArrayBase<Sq, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D>,
    ) -> Result<(), InterpolateError>
    where
        Sq: Data<Elem = Sd::Elem>,
        D: RemoveAxis,
    {
        eval_1d(&self.dev, xs, buffer, ffi::NDI_EVAL_DEFAULT)
    }

    /// `Interp1D::interp_array`'s own buffer (dropped on `Err`, src/interp1d/mod.rs:209-210): the library may skip the
    /// range pre-pass (`NDI_EVAL_FRESH_OUTPUT`); the error report is unchanged.
    fn interp_array_into_owned<Sq>(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        xs: &ArrayBase<Sq, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D>,
    ) -> Result<(), InterpolateError>
    where
        Sq: Data<Elem = Sd::Elem>,
        D: RemoveAxis,
    {
        eval_1d(&self.dev, xs, buffer, ffi::NDI_EVAL_FRESH_OUTPUT)
    }
}

// =================================================================================================
// CubicSpline (src/interp1d/strategies/cubic_spline.rs)
// =================================================================================================
/// SingleBoundary (cubic_spline.rs:204-217)
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum SingleBoundary {
    NotAKnot,
    Natural,
    Clamped,
    FirstDeriv(f64),
    SecondDeriv(f64),
}
impl SingleBoundary {
    fn to_ffi(self) -> ffi::ndi_boundary {
        let (kind, value) = match self {
            SingleBoundary::NotAKnot => (ffi::NDI_BC_NOT_A_KNOT, 0.0),
            SingleBoundary::Natural => (ffi::NDI_BC_NATURAL, 0.0),
            SingleBoundary::Clamped => (ffi::NDI_BC_CLAMPED, 0.0),
            SingleBoundary::FirstDeriv(v) => (ffi::NDI_BC_FIRST_DERIV, v),
            SingleBoundary::SecondDeriv(v) => (ffi::NDI_BC_SECOND_DERIV, v),
        };
        ffi::ndi_boundary { kind, value }
    }
}
/// RowBoundary (cubic_spline.rs:170-202)
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum RowBoundary {
    NotAKnot,
    Natural,
    Clamped,
    Mixed { left: SingleBoundary, right: SingleBoundary },
}
impl RowBoundary {
    fn ends(self) -> (SingleBoundary, SingleBoundary) {
        match self {
            RowBoundary::NotAKnot => (SingleBoundary::NotAKnot, SingleBoundary::NotAKnot),
            RowBoundary::Natural => (SingleBoundary::Natural, SingleBoundary::Natural),
            RowBoundary::Clamped => (SingleBoundary::Clamped, SingleBoundary::Clamped),
            RowBoundary::Mixed { left, right } => (left, right),
        }
    }
}
/// BoundaryCondition (cubic_spline.rs:153-168).  `Individual` holds one RowBoundary per trailing element in C order
/// with the shape `[1, data.shape()[1..]]` the reference requires (:332-340).
#[derive(Debug, Clone, PartialEq, Default)]
pub enum BoundaryCondition {
    #[default]
    NotAKnot,
    Natural,
    Clamped,
    Periodic,
    Individual { shape: Vec<usize>, rows: Vec<RowBoundary> },
}

pub(crate) struct LaneBoundaries {
    left_kind: Vec<i32>,
    left_value: Vec<f64>,
    right_kind: Vec<i32>,
    right_value: Vec<f64>,
}

#[derive(Debug, Default)]
pub struct HipCubicSpline {
    extrapolate: bool,
    boundary: BoundaryCondition,
    device: Option<i32>,
    reference_order: bool,
}
impl HipCubicSpline {
    /// default boundary NotAKnot (cubic_spline.rs:724-729)
    pub fn new() -> Self {
        Self::default()
    }
    pub fn extrapolate(mut self, extrapolate: bool) -> Self {
        self.extrapolate = extrapolate;
        self
    }
    pub fn boundary(mut self, boundary: BoundaryCondition) -> Self {
        self.boundary = boundary;
        self
    }
    pub fn device(mut self, ordinal: i32) -> Self {
        self.device = Some(ordinal);
        self
    }
    /// `NDI_BUILD_REFERENCE_ORDER` (include/ndinterp.h): never re-associate the Thomas sweeps -- the a / b tables are
    /// bit-identical to `CubicSpline::build`'s (cubic_spline.rs:678-721) for every shape, at the serial kernels'
    /// speed on narrow trailing axes with many knots.  Default: blocked sweeps there (a few ulp, see the header).
    pub fn reference_order(mut self, yes: bool) -> Self {
        self.reference_order = yes;
        self
    }
}

#[derive(Debug)]
pub struct HipCubicSplineStrategy {
    pub(crate) dev: DeviceTables1D,
}
impl HipCubicSplineStrategy {
    /// CubicSplineStrategy{a, b} (cubic_spline.rs:94-102), copied back from the device: two `(n-1) * lanes` tables.
    pub fn coefficients<T: Clone + num_traits::Zero + 'static>(&self, n: usize) -> (Vec<T>, Vec<T>) {
        assert!(dtype_of::<T>().is_some());
        let len = (n - 1) * self.dev.lanes;
        let (mut a, mut b) = (vec![T::zero(); len], vec![T::zero(); len]);
        let st = unsafe {
            ffi::ndi_interp1d_coefficients(self.dev.h, a.as_mut_ptr().cast(), b.as_mut_ptr().cast(), ffi::NDI_MEM_HOST)
        };
        assert_eq!(st, ffi::NDI_OK, "ndinterp_hip: {}", ffi::last_error());
        (a, b)
    }
}

impl<Sd, Sx, D> Interp1DStrategyBuilder<Sd, Sx, D> for HipCubicSpline
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
{
    const MINIMUM_DATA_LENGHT: usize = 3; // cubic_spline.rs:751
    type FinishedStrat = HipCubicSplineStrategy;
    fn build<Sx2>(self, x: &ArrayBase<Sx2, Ix1>, data: &ArrayBase<Sd, D>) -> Result<Self::FinishedStrat, BuilderError>
    where
        Sx2: Data<Elem = Sd::Elem>,
    {
        let device = self.device.unwrap_or_else(current_device);
        let global = |b: SingleBoundary| (false, b.to_ffi(), b.to_ffi(), None);
        let (periodic, left, right, lanes_bc) = match &self.boundary {
            BoundaryCondition::NotAKnot => global(SingleBoundary::NotAKnot),
            BoundaryCondition::Natural => global(SingleBoundary::Natural),
            BoundaryCondition::Clamped => global(SingleBoundary::Clamped),
            BoundaryCondition::Periodic => (true, ffi::ndi_boundary::default(), ffi::ndi_boundary::default(), None),
            BoundaryCondition::Individual { shape, rows } => {
                // shape must be [1, data.shape()[1..]] (cubic_spline.rs:332-340)
                let mut expect = data.shape().to_vec();
                expect[0] = 1;
                if *shape != expect {
                    return Err(BuilderError::ShapeError(format!(
                        "Boundary conditions array has wrong shape. Expected: {expect:?}, got: {shape:?}"
                    )));
                }
                let mut l = LaneBoundaries {
                    left_kind: Vec::with_capacity(rows.len()),
                    left_value: Vec::with_capacity(rows.len()),
                    right_kind: Vec::with_capacity(rows.len()),
                    right_value: Vec::with_capacity(rows.len()),
                };
                for r in rows {
                    let (lb, rb) = r.ends();
                    let (lf, rf) = (lb.to_ffi(), rb.to_ffi());
                    l.left_kind.push(lf.kind);
                    l.left_value.push(lf.value);
                    l.right_kind.push(rf.kind);
                    l.right_value.push(rf.value);
                }
                (false, ffi::ndi_boundary::default(), ffi::ndi_boundary::default(), Some(l))
            }
        };
        // Extrapolate::{Yes, No, Periodic} is decided inside the library exactly as cubic_spline.rs:763-769 does
        let dev = create_1d::<Sd::Elem, _, _, _>(
            ffi::NDI_CUBIC_SPLINE,
            self.extrapolate,
            device,
            x,
            data,
            periodic,
            if self.reference_order { ffi::NDI_BUILD_REFERENCE_ORDER } else { ffi::NDI_BUILD_DEFAULT },
            left,
            right,
            lanes_bc.as_ref(),
        )?;
        Ok(HipCubicSplineStrategy { dev })
    }
}

impl<Sd, Sx, D> Interp1DStrategy<Sd, Sx, D> for HipCubicSplineStrategy
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
{
    fn interp_into(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        target: ArrayViewMut<'_, Sd::Elem, D::Smaller>,
        x: Sx::Elem,
    ) -> Result<(), InterpolateError> {
        let xs = ndarray::arr1(&[x]);
        eval_1d(&self.dev, &xs, target.insert_axis(ndarray::Axis(0)), ffi::NDI_EVAL_DEFAULT)
    }

    fn interp_array_into<Sq>(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        xs: &ArrayBase<Sq, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D>,
    ) -> Result<(), InterpolateError>
    where
        Sq: Data<Elem = Sd::Elem>,
        D: RemoveAxis,
    {
        eval_1d(&self.dev, xs, buffer, ffi::NDI_EVAL_DEFAULT)
    }

    /// `Interp1D::interp_array`'s own buffer (dropped on `Err`, src/interp1d/mod.rs:209-210): the library may skip the
    /// range pre-pass (`NDI_EVAL_FRESH_OUTPUT`); the error report is unchanged.
    fn interp_array_into_owned<Sq>(
        &self,
        _interpolator: &Interp1D<Sd, Sx, D, Self>,
        xs: &ArrayBase<Sq, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D>,
    ) -> Result<(), InterpolateError>
    where
        Sq: Data<Elem = Sd::Elem>,
        D: RemoveAxis,
    {
        eval_1d(&self.dev, xs, buffer, ffi::NDI_EVAL_FRESH_OUTPUT)
    }
}

// =================================================================================================
// Bilinear (src/interp2d/strategies/bilinear.rs)
// =================================================================================================
#[derive(Debug, Default)]
pub struct HipBilinear {
    extrapolate: bool,
    device: Option<i32>,
}
impl HipBilinear {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn extrapolate(mut self, extrapolate: bool) -> Self {
        self.extrapolate = extrapolate;
        self
    }
    pub fn device(mut self, ordinal: i32) -> Self {
        self.device = Some(ordinal);
        self
    }
}
#[derive(Debug)]
pub struct HipBilinearStrategy {
    pub(crate) dev: DeviceTables2D,
}

impl<Sd, Sx, Sy, D> Interp2DStrategyBuilder<Sd, Sx, Sy, D> for HipBilinear
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + std::ops::Sub + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    Sy: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
    D::Smaller: RemoveAxis,
{
    const MINIMUM_DATA_LENGHT: usize = 2; // bilinear.rs:41
    type FinishedStrat = HipBilinearStrategy;
    fn build(
        self,
        x: &ArrayBase<Sx, Ix1>,
        y: &ArrayBase<Sy, Ix1>,
        data: &ArrayBase<Sd, D>,
    ) -> Result<Self::FinishedStrat, BuilderError> {
        let Some(dtype) = dtype_of::<Sd::Elem>() else {
            return Err(BuilderError::ValueError(
                "the MI355X strategies cover f32 and f64; use the generic Bilinear for other element types".into(),
            ));
        };
        let (x, y, data) = (x.as_standard_layout(), y.as_standard_layout(), data.as_standard_layout());
        let (nx, ny) = (data.shape()[0], data.shape()[1]);
        let lanes = if nx * ny == 0 { 0 } else { data.len() / (nx * ny) };
        let desc = ffi::ndi_interp2d_desc {
            dtype,
            extrapolate: self.extrapolate as i32,
            device: self.device.unwrap_or_else(current_device),
            memspace: ffi::NDI_MEM_HOST,
            nx: nx as u64,
            ny: ny as u64,
            lanes: lanes as u64,
            x_len: x.len() as u64,
            y_len: y.len() as u64,
            x: x.as_ptr().cast(),
            y: y.as_ptr().cast(),
            data: data.as_ptr().cast(),
            validate: 0, // Interp2DBuilder::build has run its checks (interp2d/mod.rs:477-509)
            reserved: 0,
        };
        let mut h = null_mut();
        match unsafe { ffi::ndi_interp2d_create(&desc, &mut h) } {
            ffi::NDI_OK => Ok(HipBilinearStrategy { dev: DeviceTables2D { h, lanes } }),
            ffi::NDI_NOT_ENOUGH_DATA => Err(BuilderError::NotEnoughData(ffi::last_error())),
            ffi::NDI_MONOTONIC => Err(BuilderError::Monotonic(ffi::last_error())),
            ffi::NDI_SHAPE => Err(BuilderError::ShapeError(ffi::last_error())),
            _ => panic!("ndinterp_hip: {}", ffi::last_error()),
        }
    }
}

/// `flags`: see `eval_1d`.
fn eval_2d<T, Sqx, Sqy, D>(
    dev: &DeviceTables2D,
    xs: &ArrayBase<Sqx, Ix1>,
    ys: &ArrayBase<Sqy, Ix1>,
    buffer: ArrayViewMut<'_, T, D>,
    flags: i32,
) -> Result<(), InterpolateError>
where
    T: Num + NumCast + Copy + Debug + 'static,
    Sqx: Data<Elem = T>,
    Sqy: Data<Elem = T>,
    D: Dimension + RemoveAxis,
{
    assert!(xs.shape() == ys.shape(), "`xs.shape()` and `ys.shape()` do not match"); // interp2d/mod.rs:189-192
    let (xs, ys) = (xs.as_standard_layout(), ys.as_standard_layout());
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_HOST,
        stream: per_thread_stream(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    let finish = |st: i32, info: &ffi::ndi_oob_info| match st {
        ffi::NDI_OK => Ok(()),
        // x is tested before y for the same query (bilinear.rs:71-80)
        ffi::NDI_OUT_OF_BOUNDS => Err(InterpolateError::OutOfBounds(oob_message::<T>(
            if info.axis == 0 { "x" } else { "y" },
            info.value,
            false,
        ))),
        ffi::NDI_NAN_QUERY => unimplemented!("failed to convert NaN to usize"),
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),
    };
    match rows_of(buffer, dev.lanes) {
        Rows::Direct { ptr, stride } => {
            let st = unsafe {
                ffi::ndi_interp2d_eval(dev.h, xs.as_ptr().cast(), ys.as_ptr().cast(), xs.len() as u64, ptr.cast(), stride as u64, &opts, &mut info)
            };
            finish(st, &info)
        }
        Rows::Bounce { mut tmp, mut dst } => {
            let st = unsafe {
                ffi::ndi_interp2d_eval(dev.h, xs.as_ptr().cast(), ys.as_ptr().cast(), xs.len() as u64, tmp.as_mut_ptr().cast(), dev.lanes as u64, &opts, &mut info)
            };
            let good = if st == ffi::NDI_OK { xs.len() } else { info.index as usize };
            dst.slice_axis_mut(ndarray::Axis(0), (0..good).into())
                .assign(&tmp.slice_axis(ndarray::Axis(0), (0..good).into()));
            finish(st, &info)
        }
    }
}

impl<Sd, Sx, Sy, D> Interp2DStrategy<Sd, Sx, Sy, D> for HipBilinearStrategy
where
    Sd: Data,
    Sd::Elem: Num + PartialOrd + NumCast + Copy + Debug + std::ops::Sub + Send + 'static,
    Sx: Data<Elem = Sd::Elem>,
    Sy: Data<Elem = Sd::Elem>,
    D: Dimension + RemoveAxis,
    D::Smaller: RemoveAxis,
{
    fn interp_into(
        &self,
        _interpolator: &Interp2D<Sd, Sx, Sy, D, Self>,
        target: ArrayViewMut<'_, Sd::Elem, <D::Smaller as Dimension>::Smaller>,
        x: Sx::Elem,
        y: Sy::Elem,
    ) -> Result<(), InterpolateError> {
        let (xs, ys) = (ndarray::arr1(&[x]), ndarray::arr1(&[y]));
        eval_2d(&self.dev, &xs, &ys, target.insert_axis(ndarray::Axis(0)), ffi::NDI_EVAL_DEFAULT)
    }

    fn interp_array_into<Sqx, Sqy>(
        &self,
        _interpolator: &Interp2D<Sd, Sx, Sy, D, Self>,
        xs: &ArrayBase<Sqx, Ix1>,
        ys: &ArrayBase<Sqy, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D::Smaller>,
    ) -> Result<(), InterpolateError>
    where
        Sqx: Data<Elem = Sd::Elem>,
        Sqy: Data<Elem = Sd::Elem>,
    {
        eval_2d(&self.dev, xs, ys, buffer, ffi::NDI_EVAL_DEFAULT)
    }

    /// `Interp2D::interp_array`'s own buffer (dropped on `Err`, src/interp2d/mod.rs:194-195): `NDI_EVAL_FRESH_OUTPUT`.
    fn interp_array_into_owned<Sqx, Sqy>(
        &self,
        _interpolator: &Interp2D<Sd, Sx, Sy, D, Self>,
        xs: &ArrayBase<Sqx, Ix1>,
        ys: &ArrayBase<Sqy, Ix1>,
        buffer: ArrayViewMut<'_, Sd::Elem, D::Smaller>,
    ) -> Result<(), InterpolateError>
    where
        Sqx: Data<Elem = Sd::Elem>,
        Sqy: Data<Elem = Sd::Elem>,
    {
        eval_2d(&self.dev, xs, ys, buffer, ffi::NDI_EVAL_FRESH_OUTPUT)
    }
}
