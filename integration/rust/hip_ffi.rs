//! Rust side of `include/ivx.h` for `datafusion-bio-function-ranges`.
//!
//! NOT compiled in this repository (the build image has no Rust toolchain): it is the file a maintainer
//! drops into `datafusion/bio-function-ranges/src/` next to a `build.rs` that links `libivx_hip.so`
//! (`println!("cargo:rustc-link-lib=dylib=ivx_hip")`).  The declarations mirror `include/ivx.h` one to one:
//! `tests/test_abi.py` parses both files and compares every function's parameter TYPES and return type through a
//! C -> Rust FFI type map, and requires every function of the header to be declared here.
#![allow(dead_code)]

use std::ffi::{c_char, c_void, CStr};

use datafusion::common::{DataFusionError, Result};

#[repr(C)] pub struct IvxCtx { _p: [u8; 0] }
#[repr(C)] pub struct IvxIndex { _p: [u8; 0] }
/// `ivx_metrics`: BuildProbeJoinMetrics under the reference's names (joins/utils.rs:399-453)
#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct IvxMetrics {
    pub build_time: f64, pub join_time: f64,
    pub build_input_batches: u64, pub build_input_rows: u64, pub build_mem_used: u64,
    pub input_batches: u64, pub input_rows: u64, pub output_batches: u64, pub output_rows: u64,
}

pub const IVX_OK: i32 = 0;
pub const IVX_ERR_OOM: i32 = 4;
pub const IVX_ERR_CAPACITY: i32 = 5;
pub const IVX_MEM_HOST: i32 = 0;
pub const IVX_MEM_DEVICE: i32 = 1;
pub const IVX_KIND_OVERLAP: i32 = 0;
pub const IVX_KIND_COUNT: i32 = 1;
pub const IVX_KIND_COVERAGE: i32 = 2;
pub const IVX_KIND_NEAREST: i32 = 3;
pub const IVX_NULL_IDX: u32 = u32::MAX;

#[link(name = "ivx_hip")]
extern "C" {
    pub fn ivx_ctx_create(device: i32, out: *mut *mut IvxCtx) -> i32;
    pub fn ivx_ctx_free(ctx: *mut IvxCtx);
    pub fn ivx_last_error(ctx: *const IvxCtx) -> *const c_char;
    pub fn ivx_ctx_set_stream(ctx: *mut IvxCtx, hip_stream: *mut c_void) -> i32;
    pub fn ivx_ctx_use_own_stream(ctx: *mut IvxCtx) -> i32;
    pub fn ivx_ctx_synchronize(ctx: *mut IvxCtx) -> i32;
    pub fn ivx_ctx_set_build_overlap(ctx: *mut IvxCtx, on: i32) -> i32;
    pub fn ivx_ctx_last_kernel_ms(ctx: *const IvxCtx) -> f64;
    pub fn ivx_version() -> *const c_char;
    pub fn ivx_ctx_metrics(ctx: *const IvxCtx, out: *mut IvxMetrics) -> i32;
    pub fn ivx_ctx_reset_metrics(ctx: *mut IvxCtx);
    pub fn ivx_ctx_set_memory_limit(ctx: *mut IvxCtx, bytes: u64) -> i32;
    pub fn ivx_ctx_reserved_bytes(ctx: *const IvxCtx) -> u64;
    pub fn ivx_ctx_trim(ctx: *mut IvxCtx, keep_bytes: u64) -> i32;
    pub fn ivx_index_build(ctx: *mut IvxCtx, kind: i32, mem: i32, key: *const u32, start: *const i32, end: *const i32,
                           n: u64, n_keys: u32, out: *mut *mut IvxIndex) -> i32;
    pub fn ivx_index_free(ix: *mut IvxIndex);
    pub fn ivx_index_rows(ix: *const IvxIndex) -> u64;
    pub fn ivx_index_device_bytes(ix: *const IvxIndex) -> u64;
    pub fn ivx_probe_overlap_count(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                                   end: *const i32, n: u64, per_row: *mut u32, total: *mut u64) -> i32;
    pub fn ivx_probe_overlap_fill(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                                  end: *const i32, n: u64, build_idx: *mut u32, probe_idx: *mut u32, cap: u64,
                                  written: *mut u64) -> i32;
    pub fn ivx_probe_exists(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                            end: *const i32, n: u64, exists: *mut u8) -> i32;
    pub fn ivx_probe_count(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                           end: *const i32, n: u64, strict: i32, out: *mut i64) -> i32;
    pub fn ivx_probe_coverage(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                              end: *const i32, n: u64, strict: i32, out: *mut i64) -> i32;
    pub fn ivx_probe_nearest(ctx: *mut IvxCtx, ix: *const IvxIndex, mem: i32, key: *const u32, start: *const i32,
                             end: *const i32, n: u64, strict: i32, k: u32, include_overlaps: i32, build_idx: *mut u32,
                             probe_idx: *mut u32, distance: *mut i64, cap: u64, rows: *mut u64) -> i32;
    pub fn ivx_merge(ctx: *mut IvxCtx, mem: i32, key: *const u32, start: *const i64, end: *const i64, n: u64, n_keys: u32,
                     min_dist: i64, strict: i32, out_key: *mut u32, out_start: *mut i64, out_end: *mut i64,
                     out_n: *mut i64, cap: u64, n_out: *mut u64) -> i32;
    pub fn ivx_subtract(ctx: *mut IvxCtx, mem: i32, lkey: *const u32, ls: *const i64, le: *const i64, nl: u64,
                        rkey: *const u32, rs: *const i64, re: *const i64, nr: u64, n_keys: u32, strict: i32,
                        out_key: *mut u32, out_start: *mut i64, out_end: *mut i64, out_row: *mut u32, cap: u64,
                        n_out: *mut u64) -> i32;
    pub fn ivx_cluster(ctx: *mut IvxCtx, mem: i32, key: *const u32, start: *const i64, end: *const i64, n: u64, n_keys: u32,
                       min_dist: i64, strict: i32, key_base: *const i64, out_key: *mut u32, out_start: *mut i64,
                       out_end: *mut i64, out_row: *mut u32, out_cluster: *mut i64, out_cluster_start: *mut i64,
                       out_cluster_end: *mut i64, key_clusters: *mut u64, n_clusters: *mut u64) -> i32;
    pub fn ivx_complement(ctx: *mut IvxCtx, mem: i32, key: *const u32, start: *const i64, end: *const i64, n: u64,
                          vkey: *const u32, vstart: *const i64, vend: *const i64, nv: u64, n_keys: u32, strict: i32,
                          out_key: *mut u32, out_start: *mut i64, out_end: *mut i64, cap: u64, n_out: *mut u64) -> i32;
    pub fn ivx_take_fixed(ctx: *mut IvxCtx, mem: i32, src: *const c_void, width: u32, n_src: u64, src_valid_bits: *const u8,
                          idx: *const u32, n: u64, out: *mut c_void, out_valid: *mut u8) -> i32;
    pub fn ivx_scatter_fixed(ctx: *mut IvxCtx, mem: i32, src: *const c_void, width: u32, idx: *const u32, n: u64,
                             out: *mut c_void, n_out: u64) -> i32;
    pub fn ivx_take_bits(ctx: *mut IvxCtx, mem: i32, src_bits: *const u8, n_src: u64, src_valid_bits: *const u8,
                         idx: *const u32, n: u64, out_bits: *mut u8, out_valid: *mut u8) -> i32;
    pub fn ivx_take_utf8(ctx: *mut IvxCtx, mem: i32, large: i32, offsets: *const c_void, data: *const u8, n_src: u64,
                         src_data_bytes: u64, src_valid_bits: *const u8, idx: *const u32, n: u64, out_offsets: *mut c_void,
                         out_data: *mut u8, data_cap: u64, data_bytes: *mut u64, out_valid: *mut u8) -> i32;
    pub fn ivx_take_view(ctx: *mut IvxCtx, mem: i32, views: *const c_void, data_bufs: *const *const u8,
                         data_buf_bytes: *const u64, n_bufs: u32, n_src: u64, src_valid_bits: *const u8, idx: *const u32,
                         n: u64, out_views: *mut c_void, out_data: *mut u8, data_cap: u64, data_bytes: *mut u64,
                         out_valid: *mut u8) -> i32;
}

/// One context per DataFusion partition stream (owns a HIP stream and scratch memory).
pub struct HipCtx(*mut IvxCtx);
unsafe impl Send for HipCtx {}

impl HipCtx {
    pub fn new(device: i32) -> Result<Self> {
        let mut p = std::ptr::null_mut();
        match unsafe { ivx_ctx_create(device, &mut p) } {
            IVX_OK => Ok(Self(p)),
            st => Err(DataFusionError::Execution(format!("no usable gfx950 device (ivx status {st})"))),
        }
    }
    fn check(&self, st: i32) -> Result<()> {
        if st == IVX_OK { return Ok(()); }
        let msg = unsafe { CStr::from_ptr(ivx_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(if st == IVX_ERR_OOM { DataFusionError::ResourcesExhausted(msg) } else { DataFusionError::Execution(msg) })
    }
}
impl Drop for HipCtx { fn drop(&mut self) { unsafe { ivx_ctx_free(self.0) } } }

impl HipCtx {
    /// Wait for every kernel this context has launched (device-resident calls may return with kernels in flight).
    pub fn synchronize(&self) -> Result<()> { self.check(unsafe { ivx_ctx_synchronize(self.0) }) }
    /// `MemoryReservation` of the partition (interval_join.rs:614-639): scratch + the indexes this context built.
    pub fn set_memory_limit(&self, bytes: u64) -> Result<()> { self.check(unsafe { ivx_ctx_set_memory_limit(self.0, bytes) }) }
    /// Where the stream ends and the reference frees its reservation: scratch back to the device.
    pub fn trim(&self, keep_bytes: u64) -> Result<()> { self.check(unsafe { ivx_ctx_trim(self.0, keep_bytes) }) }
    /// BuildProbeJoinMetrics of the calls made on this context (joins/utils.rs:399-453).
    pub fn metrics(&self) -> Result<IvxMetrics> {
        let mut m = IvxMetrics::default();
        self.check(unsafe { ivx_ctx_metrics(self.0, &mut m) })?;
        Ok(m)
    }
}

/// Immutable after build: shared by every probe stream through `Arc<JoinLeftData>` like today's trees.
/// `probers`: the contexts that probed it with device-resident buffers; `ivx_index_free`'s contract
/// (include/ivx.h) wants each of them synchronised before the index's buffers go back to the pool.
pub struct HipIndex { ix: *mut IvxIndex, probers: std::sync::Mutex<Vec<std::sync::Arc<HipCtx>>> }
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {}
unsafe impl Sync for HipCtx {}
impl HipIndex {
    /// A partition stream registers its context before its first device-resident probe.
    pub fn register_prober(&self, ctx: std::sync::Arc<HipCtx>) { self.probers.lock().unwrap().push(ctx); }
}
impl Drop for HipIndex {
    fn drop(&mut self) {
        for c in self.probers.lock().unwrap().iter() { let _ = unsafe { ivx_ctx_synchronize(c.0) }; }
        unsafe { ivx_index_free(self.ix) }
    }
}

impl HipCtx {
    /// `collect_left_input` (interval_join.rs:641-662): key ids from the join's key dictionary, Int32 coordinates.
    pub fn build_overlap(&self, key_ids: &[u32], start: &[i32], end: &[i32], n_keys: u32) -> Result<HipIndex> {
        let mut ix = std::ptr::null_mut();
        self.check(unsafe { ivx_index_build(self.0, IVX_KIND_OVERLAP, IVX_MEM_HOST, key_ids.as_ptr(), start.as_ptr(),
                                            end.as_ptr(), start.len() as u64, n_keys, &mut ix) })?;
        Ok(HipIndex { ix, probers: Default::default() })
    }

    /// `process_probe_batch` full mode (interval_join.rs:1614-1653): the (left_indexes, index_right) arrays fed to
    /// `compute::take`.  Counting first sizes the buffers exactly and gives the library its density hint.
    pub fn probe_overlap(&self, ix: &HipIndex, key_ids: &[u32], start: &[i32], end: &[i32]) -> Result<(Vec<u32>, Vec<u32>)> {
        let n = start.len() as u64;
        let mut total = 0u64;
        self.check(unsafe { ivx_probe_overlap_count(self.0, ix.ix, IVX_MEM_HOST, key_ids.as_ptr(), start.as_ptr(), end.as_ptr(),
                                                    n, std::ptr::null_mut(), &mut total) })?;
        let (mut b, mut p) = (vec![0u32; total as usize], vec![0u32; total as usize]);
        let mut written = 0u64;
        self.check(unsafe { ivx_probe_overlap_fill(self.0, ix.ix, IVX_MEM_HOST, key_ids.as_ptr(), start.as_ptr(), end.as_ptr(),
                                                   n, b.as_mut_ptr(), p.as_mut_ptr(), total, &mut written) })?;
        b.truncate(written as usize); p.truncate(written as usize);
        Ok((b, p))
    }
}
