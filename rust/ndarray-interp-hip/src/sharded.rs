//! One call, several devices.  The reference is single-threaded; its multi-worker shape is one interpolator
//! driven from many threads over contiguous blocks of the query array (benches/bench_interp1d.rs:49-79).
//! `ndi_interp1d_eval_sharded` is that shape below the host language: N replica handles (one per device, knots /
//! data / coefficients replicated), one flattened query array, one host thread per device inside the library, and the
//! serial loop's first-error result over the WHOLE batch (src/interp1d/mod.rs:326-343): every shard range-checks
//! its block, the shards agree on the minimum failing flat index F, exactly the rows [0, F) are produced.
//! No device-to-device traffic, no collective.
use std::fmt::Debug;
use std::ptr::{null, null_mut};

use ndarray::{ArrayBase, ArrayViewMut, Data, Dimension, Ix1, RemoveAxis};
use ndarray_interp::InterpolateError;
use num_traits::NumCast;

use crate::hip_ffi as ffi;
use crate::strategies::{DeviceTables1D, DeviceTables2D};

/// A replica of built device tables on `device` (tables copied device to device -- over xGMI between two GPUs;
/// nothing is uploaded or solved again).
pub fn replicate_1d(src: &DeviceTables1D, device: i32) -> DeviceTables1D {
    let mut h = null_mut();
    let st = unsafe { ffi::ndi_interp1d_clone(src.h, device, &mut h) };
    assert_eq!(st, ffi::NDI_OK, "ndinterp_hip: {}", ffi::last_error());
    DeviceTables1D { h, lanes: src.lanes }
}
pub fn replicate_2d(src: &DeviceTables2D, device: i32) -> DeviceTables2D {
    let mut h = null_mut();
    let st = unsafe { ffi::ndi_interp2d_clone(src.h, device, &mut h) };
    assert_eq!(st, ffi::NDI_OK, "ndinterp_hip: {}", ffi::last_error());
    DeviceTables2D { h, lanes: src.lanes }
}

/// Contiguous block of shard `i` of `n`: sizes differ by at most one (`ndi_shard_bounds`).
pub fn shard_bounds(nq: usize, i: u32, n: u32) -> (usize, usize) {
    let (mut lo, mut hi) = (0u64, 0u64);
    unsafe { ffi::ndi_shard_bounds(nq as u64, i, n, &mut lo, &mut hi) };
    (lo as usize, hi as usize)
}

/// `interp_array_into` over every replica: contiguous blocks of `xs` per replica, rows land in `buffer`
/// (host arrays in and out; `buffer` must be C-contiguous: the shards write straight into it).
pub fn interp_array_into_sharded_1d<T, Sq, D>(
    replicas: &[&DeviceTables1D],
    xs: &ArrayBase<Sq, Ix1>,
    mut buffer: ArrayViewMut<'_, T, D>,
) -> Result<(), InterpolateError>
where
    T: NumCast + Copy + Debug + 'static,
    Sq: Data<Elem = T>,
    D: Dimension + RemoveAxis,
{
    let xs = xs.as_standard_layout();
    let n = replicas.len() as u32;
    assert!(n >= 1 && buffer.shape()[0] == xs.len());
    assert!(buffer.is_standard_layout(), "sharded evaluation writes straight into the buffer: pass a C-contiguous view");
    let lanes = replicas[0].lanes;
    let handles: Vec<*const ffi::ndi_interp1d> = replicas.iter().map(|r| r.h as *const _).collect();
    let base = buffer.as_mut_ptr();
    let io: Vec<ffi::ndi_shard_io> = (0..n)
        .map(|i| {
            let (lo, _) = shard_bounds(xs.len(), i, n);
            ffi::ndi_shard_io {
                q: null(), // the shard reads its block of `xs`
                qy: null(),
                out: unsafe { base.add(lo * lanes) }.cast(),
                stream: null_mut(),
            }
        })
        .collect();
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_HOST,
        stream: null_mut(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags: ffi::NDI_EVAL_DEFAULT,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    let st = unsafe {
        ffi::ndi_interp1d_eval_sharded(
            handles.as_ptr(),
            n,
            xs.as_ptr().cast(),
            xs.len() as u64,
            io.as_ptr(),
            lanes as u64,
            &opts,
            &mut info,
        )
    };
    match st {
        ffi::NDI_OK => Ok(()),
        ffi::NDI_OUT_OF_BOUNDS => {
            // info.index is the flat index in the WHOLE batch
            let v: T = num_traits::cast(info.value).expect("query value");
            Err(InterpolateError::OutOfBounds(format!("x = {v:#?} is not in range")))
        }
        ffi::NDI_NAN_QUERY => unimplemented!("failed to convert NaN to usize"),
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),
    }
}

/// 2-D counterpart: x is tested before y for the same query (bilinear.rs:71-80), across all shards.
pub fn interp_array_into_sharded_2d<T, Sqx, Sqy, D>(
    replicas: &[&DeviceTables2D],
    xs: &ArrayBase<Sqx, Ix1>,
    ys: &ArrayBase<Sqy, Ix1>,
    mut buffer: ArrayViewMut<'_, T, D>,
) -> Result<(), InterpolateError>
where
    T: NumCast + Copy + Debug + 'static,
    Sqx: Data<Elem = T>,
    Sqy: Data<Elem = T>,
    D: Dimension + RemoveAxis,
{
    assert!(xs.shape() == ys.shape(), "`xs.shape()` and `ys.shape()` do not match");
    let (xs, ys) = (xs.as_standard_layout(), ys.as_standard_layout());
    let n = replicas.len() as u32;
    assert!(n >= 1 && buffer.shape()[0] == xs.len() && buffer.is_standard_layout());
    let lanes = replicas[0].lanes;
    let handles: Vec<*const ffi::ndi_interp2d> = replicas.iter().map(|r| r.h as *const _).collect();
    let base = buffer.as_mut_ptr();
    let io: Vec<ffi::ndi_shard_io> = (0..n)
        .map(|i| {
            let (lo, _) = shard_bounds(xs.len(), i, n);
            ffi::ndi_shard_io { q: null(), qy: null(), out: unsafe { base.add(lo * lanes) }.cast(), stream: null_mut() }
        })
        .collect();
    let opts = ffi::ndi_eval_opts {
        q_memspace: ffi::NDI_MEM_HOST,
        out_memspace: ffi::NDI_MEM_HOST,
        stream: null_mut(),
        path: ffi::NDI_PATH_AUTO,
        async_launch: 0,
        flags: ffi::NDI_EVAL_DEFAULT,
        reserved: 0,
    };
    let mut info = ffi::ndi_oob_info::default();
    let st = unsafe {
        ffi::ndi_interp2d_eval_sharded(
            handles.as_ptr(),
            n,
            xs.as_ptr().cast(),
            ys.as_ptr().cast(),
            xs.len() as u64,
            io.as_ptr(),
            lanes as u64,
            &opts,
            &mut info,
        )
    };
    match st {
        ffi::NDI_OK => Ok(()),
        ffi::NDI_OUT_OF_BOUNDS => {
            let v: T = num_traits::cast(info.value).expect("query value");
            let axis = if info.axis == 0 { "x" } else { "y" };
            Err(InterpolateError::OutOfBounds(format!("{axis} = {v:?} is not in range")))
        }
        ffi::NDI_NAN_QUERY => unimplemented!("failed to convert NaN to usize"),
        _ => panic!("ndinterp_hip: {}", ffi::last_error()),
    }
}
