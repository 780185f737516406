//! `Interp1D::interp_array` / `Interp2D::interp_array` for outputs that do not fit (or need not stay) in device
//! memory: the reference allocates `zeros(xs.shape ++ lanes)` (src/interp1d/mod.rs:197-211) -- 327.7 GB at
//! 4096 lanes x 1e7 f64 queries, more than the 288 GB of one MI355X.  `ndi_interp{1,2}d_eval_ring` evaluates the
//! flattened queries in chunks into a ring of device buffers and hands every chunk to a consumer.
use std::fmt::Debug;
use std::os::raw::c_void;
use std::ptr::null;

use ndarray::{ArrayBase, Data, Dimension};
use ndarray_interp::InterpolateError;
use num_traits::NumCast;

use crate::hip_ffi as ffi;
use crate::strategies::{per_thread_stream, DeviceTables1D, DeviceTables2D};

/// What the consumer sees: rows `[q_begin, q_begin + q_count)` as `T[q_count][row_stride]` in device memory,
/// produced on `stream`.  Return null when the consumer's work is enqueued on `chunk.stream`; return the
/// `hipEvent_t` recorded on the consumer's own stream otherwise (the library waits for it before the slot is reused).
pub type Chunk = ffi::ndi_ring_chunk;

/// The consumer closure plus the payload of a panic it raised.  A panic must not unwind across the `extern "C"`
/// frame of the library (undefined behaviour / process abort): the trampoline catches it, parks the payload here, stops
/// calling the closure for the remaining chunks (the library keeps producing them into the ring; nobody looks at them)
/// and the panic is resumed on the Rust side once `ndi_interp{1,2}d_eval_ring` has returned.
struct Bridge<F> {
    consume: F,
    panic: Option<Box<dyn std::any::Any + Send + 'static>>,
}

unsafe extern "C" fn trampoline<F: FnMut(&Chunk) -> *mut c_void>(user: *mut c_void, c: *const Chunk) -> *mut c_void {
    let bridge = &mut *(user as *mut Bridge<F>);
    if bridge.panic.is_some() {
        return std::ptr::null_mut();
    }
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| (bridge.consume)(&*c))) {
        Ok(ev) => ev,
        Err(payload) => {
            bridge.panic = Some(payload);
            std::ptr::null_mut()
        }
    }
}

fn finish<T: NumCast + Debug>(st: i32, info: &ffi::ndi_oob_info, pretty: bool) -> Result<(), InterpolateError> {
    match st {
        ffi::NDI_OK => Ok(()),
        ffi::NDI_OUT_OF_BOUNDS => {
            let v: T = num_traits::cast(info.value).expect("query value");
            let axis = if info.axis == 0 { "x" } else { "y" };
            Err(InterpolateError::OutOfBounds(if pretty {
                format!("{axis} = {v:#?} is not in range")
            } else {
                format!("{axis} = {v:?} is not in range")
            }))
        }
        ffi::NDI_NAN_QUERY => unimplemented!("failed to convert NaN to usize"), // vector_extensions.rs:83-84
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),
    }
}

/// 1-D: `xs` of any rank is flattened in C order (what the reference's general-rank branch amounts to,
/// src/interp1d/mod.rs:301-322).  `n_slots` ring slots of `chunk` rows each; the library owns the ring
/// (one allocation, slots interleaved row by row: `chunk.row_stride == n_slots * lanes`).
/// On `Err` exactly the rows before the first failing query have been handed out (interp1d/mod.rs:334-342).
pub fn interp_array_chunks_1d<T, Sq, Dq, F>(
    dev: &DeviceTables1D,
    xs: &ArrayBase<Sq, Dq>,
    chunk: usize,
    n_slots: u32,
    consume: F,
) -> Result<(), InterpolateError>
where
    T: NumCast + Copy + Debug + 'static,
    Sq: Data<Elem = T>,
    Dq: Dimension,
    F: FnMut(&Chunk) -> *mut c_void,
{
    let xs = xs.as_standard_layout();
    let ring = ffi::ndi_ring_desc { slots: null(), n_slots, reserved: 0, chunk_queries: chunk as u64, row_stride: 0 };
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_DEVICE,
        stream: per_thread_stream(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags: ffi::NDI_EVAL_DEFAULT,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    let mut bridge = Bridge { consume, panic: None };
    let st = unsafe {
        ffi::ndi_interp1d_eval_ring(
            dev.h,
            xs.as_ptr().cast(),
            xs.len() as u64,
            &ring,
            Some(trampoline::<F>),
            &mut bridge as *mut Bridge<F> as *mut c_void,
            &opts,
            &mut info,
        )
    };
    if let Some(payload) = bridge.panic.take() {
        std::panic::resume_unwind(payload); // the consumer's panic, outside the C frames
    }
    finish::<T>(st, &info, true)
}

/// 2-D counterpart (src/interp2d/mod.rs:175-196, 287-307).
pub fn interp_array_chunks_2d<T, Sqx, Sqy, Dq, F>(
    dev: &DeviceTables2D,
    xs: &ArrayBase<Sqx, Dq>,
    ys: &ArrayBase<Sqy, Dq>,
    chunk: usize,
    n_slots: u32,
    consume: F,
) -> Result<(), InterpolateError>
where
    T: NumCast + Copy + Debug + 'static,
    Sqx: Data<Elem = T>,
    Sqy: Data<Elem = T>,
    Dq: Dimension,
    F: FnMut(&Chunk) -> *mut c_void,
{
    assert!(xs.shape() == ys.shape(), "`xs.shape()` and `ys.shape()` do not match");
    let (xs, ys) = (xs.as_standard_layout(), ys.as_standard_layout());
    let ring = ffi::ndi_ring_desc { slots: null(), n_slots, reserved: 0, chunk_queries: chunk as u64, row_stride: 0 };
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_DEVICE,
        stream: per_thread_stream(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags: ffi::NDI_EVAL_DEFAULT,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    let mut bridge = Bridge { consume, panic: None };
    let st = unsafe {
        ffi::ndi_interp2d_eval_ring(
            dev.h,
            xs.as_ptr().cast(),
            ys.as_ptr().cast(),
            xs.len() as u64,
            &ring,
            Some(trampoline::<F>),
            &mut bridge as *mut Bridge<F> as *mut c_void,
            &opts,
            &mut info,
        )
    };
    if let Some(payload) = bridge.panic.take() {
        std::panic::resume_unwind(payload);
    }
    finish::<T>(st, &info, false)
}
