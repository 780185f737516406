//! `extern "C"` bindings of libndinterp_hip.so -- one to one with `include/ndinterp.h` (v0.5).
//!
//! Every `#[repr(C)]` struct and every function below is compared with the header by
//! `tests/test_rust_ffi_abi.py`: field / argument ORDER and C TYPE, not only names.  Keep one field
//! or argument per `name: type` pair and do not reorder.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

// ---- enums (plain i32 on the ABI) -------------------------------------------------------------
/// `ndi_status`: BuilderError / InterpolateError (src/lib.rs:127-146) + ABI-only codes.
pub const NDI_OK: i32 = 0;
pub const NDI_NOT_ENOUGH_DATA: i32 = 1;
pub const NDI_MONOTONIC: i32 = 2;
pub const NDI_SHAPE: i32 = 3;
pub const NDI_VALUE: i32 = 4;
pub const NDI_OUT_OF_BOUNDS: i32 = 5;
pub const NDI_NAN_QUERY: i32 = 6;
pub const NDI_HIP_ERROR: i32 = 7;
pub const NDI_BAD_ARG: i32 = 8;
pub const NDI_UNSUPPORTED: i32 = 9;

/// `ndi_dtype`
pub const NDI_F32: i32 = 0;
pub const NDI_F64: i32 = 1;
/// `ndi_memspace`
pub const NDI_MEM_HOST: i32 = 0;
pub const NDI_MEM_DEVICE: i32 = 1;
/// `ndi_strategy1d`
pub const NDI_LINEAR: i32 = 0;
pub const NDI_CUBIC_SPLINE: i32 = 1;
/// `ndi_bc_kind`: SingleBoundary (cubic_spline.rs:204-217)
pub const NDI_BC_NOT_A_KNOT: i32 = 0;
pub const NDI_BC_NATURAL: i32 = 1;
pub const NDI_BC_CLAMPED: i32 = 2;
pub const NDI_BC_FIRST_DERIV: i32 = 3;
pub const NDI_BC_SECOND_DERIV: i32 = 4;
/// `ndi_monotonic`: Monotonic (src/vector_extensions.rs:25-29)
pub const NDI_MONO_NOT: i32 = 0;
pub const NDI_MONO_RISING_STRICT: i32 = 1;
pub const NDI_MONO_RISING: i32 = 2;
pub const NDI_MONO_FALLING_STRICT: i32 = 3;
pub const NDI_MONO_FALLING: i32 = 4;
/// `ndi_build_flags` (`ndi_interp1d_desc::build_flags`)
pub const NDI_BUILD_DEFAULT: i32 = 0;
pub const NDI_BUILD_REFERENCE_ORDER: i32 = 1;
/// `ndi_eval_flags` (`ndi_eval_opts::flags`)
pub const NDI_EVAL_DEFAULT: i32 = 0;
pub const NDI_EVAL_FRESH_OUTPUT: i32 = 1;
pub const NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED: i32 = 2;
/// `ndi_output_flags` (`ndi_output_alloc`)
pub const NDI_OUTPUT_ZEROED: i32 = 0;
pub const NDI_OUTPUT_UNINITIALIZED: i32 = 1;
/// `ndi_path`
pub const NDI_PATH_AUTO: i32 = 0;
pub const NDI_PATH_GATHER: i32 = 1;
pub const NDI_PATH_BUCKETED: i32 = 2;

// ---- structs ----------------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ndi_boundary {
    pub kind: i32,
    pub value: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_interp1d_desc {
    pub dtype: i32,
    pub strategy: i32,
    pub extrapolate: i32,
    pub device: i32,
    pub n: u64,
    pub lanes: u64,
    pub x_len: u64,
    pub x: *const c_void,
    pub data: *const c_void,
    pub memspace: i32,
    pub validate: i32,
    pub periodic: i32,
    pub build_flags: i32,
    pub left: ndi_boundary,
    pub right: ndi_boundary,
    pub lane_left_kind: *const i32,
    pub lane_left_value: *const f64,
    pub lane_right_kind: *const i32,
    pub lane_right_value: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_interp2d_desc {
    pub dtype: i32,
    pub extrapolate: i32,
    pub device: i32,
    pub memspace: i32,
    pub nx: u64,
    pub ny: u64,
    pub lanes: u64,
    pub x_len: u64,
    pub y_len: u64,
    pub x: *const c_void,
    pub y: *const c_void,
    pub data: *const c_void,
    pub validate: i32,
    pub reserved: i32,
}

/// Opaque handles (own the device copies of x, data and the spline tables).
#[repr(C)]
pub struct ndi_interp1d {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ndi_interp2d {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ndi_locator {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ndi_oob_info {
    pub index: u64,
    pub value: f64,
    pub axis: i32,
    pub status: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_eval_opts {
    pub q_memspace: i32,
    pub out_memspace: i32,
    pub stream: *mut c_void,
    pub path: i32,
    pub async_launch: i32,
    pub flags: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_ring_chunk {
    pub index: u64,
    pub q_begin: u64,
    pub q_count: u64,
    pub out: *mut c_void,
    pub row_stride: u64,
    pub slot: u32,
    pub shard: u32,
    pub stream: *mut c_void,
}

/// `void* (*ndi_ring_consumer)(void* user, const ndi_ring_chunk* chunk)`
pub type ndi_ring_consumer =
    Option<unsafe extern "C" fn(user: *mut c_void, chunk: *const ndi_ring_chunk) -> *mut c_void>;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_ring_desc {
    pub slots: *const *mut c_void,
    pub n_slots: u32,
    pub reserved: u32,
    pub chunk_queries: u64,
    pub row_stride: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ndi_shard_io {
    pub q: *const c_void,
    pub qy: *const c_void,
    pub out: *mut c_void,
    pub stream: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ndi_profile {
    pub eval_launches: u64,
    pub eval_ms: f64,
    pub locate_launches: u64,
    pub locate_ms: f64,
    pub group_launches: u64,
    pub group_ms: f64,
    pub last_path: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ndi_output_info {
    pub tries: u32,
    pub reserved: u32,
    pub fill_tbps: f64,
    pub worst_fill_tbps: f64,
    pub alloc_ms: f64,
}

// ---- functions --------------------------------------------------------------------------------
extern "C" {
    pub fn ndi_interp1d_create(desc: *const ndi_interp1d_desc, out: *mut *mut ndi_interp1d) -> i32;
    pub fn ndi_interp1d_destroy(h: *mut ndi_interp1d);
    pub fn ndi_interp2d_create(desc: *const ndi_interp2d_desc, out: *mut *mut ndi_interp2d) -> i32;
    pub fn ndi_interp2d_destroy(h: *mut ndi_interp2d);
    pub fn ndi_interp1d_clone(h: *const ndi_interp1d, device: i32, out: *mut *mut ndi_interp1d) -> i32;
    pub fn ndi_interp2d_clone(h: *const ndi_interp2d, device: i32, out: *mut *mut ndi_interp2d) -> i32;
    pub fn ndi_interp1d_coefficients(
        h: *const ndi_interp1d,
        a_out: *mut c_void,
        b_out: *mut c_void,
        memspace: i32,
    ) -> i32;
    pub fn ndi_interp1d_eval(
        h: *const ndi_interp1d,
        q: *const c_void,
        nq: u64,
        out: *mut c_void,
        out_row_stride: u64,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp2d_eval(
        h: *const ndi_interp2d,
        qx: *const c_void,
        qy: *const c_void,
        nq: u64,
        out: *mut c_void,
        out_row_stride: u64,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp1d_finish(h: *const ndi_interp1d, stream: *mut c_void, info: *mut ndi_oob_info) -> i32;
    pub fn ndi_interp2d_finish(h: *const ndi_interp2d, stream: *mut c_void, info: *mut ndi_oob_info) -> i32;
    pub fn ndi_interp1d_eval_ring(
        h: *const ndi_interp1d,
        q: *const c_void,
        nq: u64,
        ring: *const ndi_ring_desc,
        consume: ndi_ring_consumer,
        user: *mut c_void,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp2d_eval_ring(
        h: *const ndi_interp2d,
        qx: *const c_void,
        qy: *const c_void,
        nq: u64,
        ring: *const ndi_ring_desc,
        consume: ndi_ring_consumer,
        user: *mut c_void,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_shard_bounds(nq: u64, shard: u32, n_shards: u32, lo: *mut u64, hi: *mut u64);
    pub fn ndi_interp1d_eval_sharded(
        handles: *const *const ndi_interp1d,
        n_shards: u32,
        q: *const c_void,
        nq: u64,
        io: *const ndi_shard_io,
        out_row_stride: u64,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp2d_eval_sharded(
        handles: *const *const ndi_interp2d,
        n_shards: u32,
        qx: *const c_void,
        qy: *const c_void,
        nq: u64,
        io: *const ndi_shard_io,
        out_row_stride: u64,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp1d_eval_ring_sharded(
        handles: *const *const ndi_interp1d,
        n_shards: u32,
        q: *const c_void,
        nq: u64,
        io: *const ndi_shard_io,
        rings: *const ndi_ring_desc,
        consume: ndi_ring_consumer,
        user: *mut c_void,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp2d_eval_ring_sharded(
        handles: *const *const ndi_interp2d,
        n_shards: u32,
        qx: *const c_void,
        qy: *const c_void,
        nq: u64,
        io: *const ndi_shard_io,
        rings: *const ndi_ring_desc,
        consume: ndi_ring_consumer,
        user: *mut c_void,
        opts: *const ndi_eval_opts,
        info: *mut ndi_oob_info,
    ) -> i32;
    pub fn ndi_interp1d_trim(h: *const ndi_interp1d) -> i32;
    pub fn ndi_interp2d_trim(h: *const ndi_interp2d) -> i32;
    pub fn ndi_interp1d_scratch_sets(h: *const ndi_interp1d) -> u64;
    pub fn ndi_get_lower_index_batch(
        dtype: i32,
        device: i32,
        knots: *const c_void,
        n: u64,
        q: *const c_void,
        nq: u64,
        out_idx: *mut i64,
        memspace: i32,
    ) -> i32;
    pub fn ndi_locator_create(
        dtype: i32,
        device: i32,
        knots: *const c_void,
        n: u64,
        memspace: i32,
        out: *mut *mut ndi_locator,
    ) -> i32;
    pub fn ndi_locator_eval(
        h: *const ndi_locator,
        q: *const c_void,
        nq: u64,
        out_idx: *mut i64,
        memspace: i32,
        stream: *mut c_void,
    ) -> i32;
    pub fn ndi_locator_destroy(h: *mut ndi_locator);
    pub fn ndi_monotonic_prop(dtype: i32, host_v: *const c_void, n: u64) -> i32;
    pub fn ndi_validate1d(dtype: i32, host_x: *const c_void, x_len: u64, n: u64, strategy: i32) -> i32;
    pub fn ndi_validate2d(
        dtype: i32,
        host_x: *const c_void,
        x_len: u64,
        host_y: *const c_void,
        y_len: u64,
        nx: u64,
        ny: u64,
    ) -> i32;
    pub fn ndi_output_alloc(
        device: i32,
        bytes: u64,
        max_tries: u32,
        flags: u32,
        out: *mut *mut c_void,
        info: *mut ndi_output_info,
    ) -> i32;
    pub fn ndi_output_free(p: *mut c_void) -> i32;
    pub fn ndi_output_trim() -> i32;
    pub fn ndi_device_count() -> i32;
    pub fn ndi_last_error_string() -> *const c_char;
    pub fn ndi_version() -> u32;
    pub fn ndi_profile_enable(on: i32);
    pub fn ndi_profile_read(out: *mut ndi_profile, reset: i32) -> i32;
    pub fn ndi_interp2d_probe_ceiling(
        h: *const ndi_interp2d,
        nq: u64,
        out: *mut c_void,
        out_row_stride: u64,
        stream: *mut c_void,
        reps: i32,
        ms: *mut f64,
    ) -> i32;
}

/// The calling thread's last error text (`ndi_last_error_string`).
pub fn last_error() -> String {
    // Safety: the library returns a NUL-terminated string owned by a thread-local that lives until the
    // thread's next failing call.
    unsafe { std::ffi::CStr::from_ptr(ndi_last_error_string()).to_string_lossy().into_owned() }
}
