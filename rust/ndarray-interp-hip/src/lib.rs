//! MI355X (gfx950) backend for the batched `interp_array` path of `ndarray-interp` 0.6.0.
//!
//! * [`hip_ffi`] -- the `extern "C"` block, one to one with `include/ndinterp.h`.
//! * [`strategies`] -- `HipLinear`, `HipCubicSpline`, `HipBilinear`: strategy builders for the reference's
//!   `Interp1DBuilder` / `Interp2DBuilder`, whose finished strategies override the batched trait hook
//!   (`rust/patches/ndarray-interp-0.6.0-batched-hook.patch`) with one C-ABI call.
//! * [`ring`] -- `interp_array` in chunks through a device-output ring, for outputs larger than HBM.
//! * [`sharded`] -- one call over several devices (one replica handle per device).
//!
//! Uncompiled in the repository that carries it (no Rust toolchain in that image); the C ABI underneath is
//! exercised by that repository's ctypes / C++ / C99 tests, and `tests/test_rust_ffi_abi.py` checks this crate's
//! declarations against the header mechanically.
pub mod hip_ffi;
pub mod ring;
pub mod sharded;
pub mod strategies;

pub use strategies::{
    current_device, set_current_device, BoundaryCondition, HipBilinear, HipBilinearStrategy, HipCubicSpline,
    HipCubicSplineStrategy, HipLinear, HipLinearStrategy, RowBoundary, SingleBoundary,
};

/// `use ndarray_interp_hip::prelude::*;` after the reference's own imports swaps the built-in strategy names for the
/// device ones, so existing builder chains (`.strategy(Linear::new())`) compile unchanged.
pub mod prelude {
    pub use crate::strategies::HipBilinear as Bilinear;
    pub use crate::strategies::HipCubicSpline as CubicSpline;
    pub use crate::strategies::HipLinear as Linear;
    pub use crate::strategies::{BoundaryCondition, RowBoundary, SingleBoundary};
}
