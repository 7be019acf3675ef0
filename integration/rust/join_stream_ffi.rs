//! Rust side of the push-style join stream of `include/bio_ranges_host.h` (`brh_join_stream_*`), the piece of
//! `IntervalJoinStream` (interval_join.rs:934-1140) that coalesces probe RecordBatches before they go to the GPU.
//!
//! NOT compiled in this repository (no Rust toolchain in the build image).  The declarations mirror the header one
//! to one (`tests/test_abi.py` checks names and parameter counts); `poll_next_impl` below shows how the reference's
//! state machine maps onto them.  Arrays cross as Arrow C Data Interface structs (`arrow::ffi`).
#![allow(dead_code)]

use std::ffi::{c_char, c_int, CStr};

use arrow::array::{ArrayRef, Int64Array, RecordBatch, StructArray, UInt32Array};
use arrow::compute::concat_batches;
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use datafusion::common::{DataFusionError, Result};

#[repr(C)] pub struct BrhSession { _p: [u8; 0] }
#[repr(C)] pub struct BrhJoinStream { _p: [u8; 0] }
#[repr(C)] pub struct BrhBatch { pub array: *const FFI_ArrowArray, pub schema: *const FFI_ArrowSchema }
#[repr(C)] pub struct BrhColumns { pub keys: *const *const c_char, pub n_keys: c_int, pub start: *const c_char, pub end: *const c_char }

#[link(name = "bio_ranges_hip")]
extern "C" {
    pub fn brh_session_create(device_ordinal: c_int, out: *mut *mut BrhSession) -> c_int;
    pub fn brh_session_free(s: *mut BrhSession);
    pub fn brh_last_error(s: *const BrhSession) -> *const c_char;
    pub fn brh_join_stream_open(s: *mut BrhSession, build: BrhBatch, bcols: BrhColumns, pcols: BrhColumns, strict_predicate: c_int,
                                coalesce_rows: u64, join_type: c_int, max_output_rows: u64, out: *mut *mut BrhJoinStream) -> c_int;
    pub fn brh_join_stream_push(js: *mut BrhJoinStream, probe: BrhBatch, n_ready: *mut c_int) -> c_int;
    pub fn brh_join_stream_finish(js: *mut BrhJoinStream, n_ready: *mut c_int) -> c_int;
    pub fn brh_join_stream_next(js: *mut BrhJoinStream, first_batch: *mut u64, n_batches: *mut u64, group_done: *mut c_int,
                                build_idx: *mut FFI_ArrowArray, build_idx_schema: *mut FFI_ArrowSchema,
                                probe_idx: *mut FFI_ArrowArray, probe_idx_schema: *mut FFI_ArrowSchema,
                                batch_offsets: *mut FFI_ArrowArray, batch_offsets_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn brh_join_stream_close(js: *mut BrhJoinStream);
}

/// What `IntervalJoinStream` keeps per partition when `Algorithm::Hip` is selected.
pub struct HipJoinStream {
    session: *mut BrhSession,
    js: *mut BrhJoinStream,
    /// probe batches pushed but not yet returned in a result (the payload `take` needs them)
    pending: Vec<RecordBatch>,
}

impl HipJoinStream {
    fn check(&self, rc: c_int) -> Result<()> {
        if rc == 0 { return Ok(()); }
        let msg = unsafe { CStr::from_ptr(brh_last_error(self.session)) }.to_string_lossy().into_owned();
        Err(DataFusionError::Execution(msg))
    }

    /// FetchProbeBatch + ProcessProbeBatch (interval_join.rs:1107-1140, :1614-1653): hand the batch over; if a
    /// coalesced group became ready, return its pairs together with the concatenated probe batches they refer to.
    pub fn push(&mut self, batch: RecordBatch) -> Result<Option<(UInt32Array, UInt32Array, RecordBatch)>> {
        let (arr, sch) = to_ffi(&StructArray::from(batch.clone()).into_data())?;
        let mut ready: c_int = 0;
        self.pending.push(batch);
        self.check(unsafe { brh_join_stream_push(self.js, BrhBatch { array: &arr, schema: &sch }, &mut ready) })?;
        if ready == 0 { Ok(None) } else { self.next().map(Some) }
    }

    /// ExhaustedProbeSide: flush the last, partial group.
    pub fn finish(&mut self) -> Result<Option<(UInt32Array, UInt32Array, RecordBatch)>> {
        let mut ready: c_int = 0;
        self.check(unsafe { brh_join_stream_finish(self.js, &mut ready) })?;
        if ready == 0 { Ok(None) } else { self.next().map(Some) }
    }

    fn next(&mut self) -> Result<(UInt32Array, UInt32Array, RecordBatch)> {
        let (mut first, mut nb, mut done) = (0u64, 0u64, 0 as c_int);
        let (mut ba, mut bs) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        let (mut pa, mut ps) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        self.check(unsafe { brh_join_stream_next(self.js, &mut first, &mut nb, &mut done, &mut ba, &mut bs, &mut pa, &mut ps, &mut oa, &mut os) })?;
        let build_idx = UInt32Array::from(unsafe { from_ffi(ba, &bs) }?);
        let probe_idx = UInt32Array::from(unsafe { from_ffi(pa, &ps) }?);
        let _offsets = Int64Array::from(unsafe { from_ffi(oa, &os) }?);   // first row of every batch in the group
        // the group's batches in push order = the rows probe_idx counts over.  A bounded-output stream
        // (max_output_rows > 0, the low-memory mode) returns several results per group: the batches go with the last one
        let group: Vec<RecordBatch> = if done != 0 { self.pending.drain(..nb as usize).collect() } else { self.pending[..nb as usize].to_vec() };
        let probe = concat_batches(&group[0].schema(), &group)?;
        Ok((build_idx, probe_idx, probe))
        // caller: build_batch_from_indices(schema, build_side, &probe, &build_idx, &probe_idx, ..)  (interval_join.rs:1655-1667)
    }
}

impl Drop for HipJoinStream {
    fn drop(&mut self) { unsafe { brh_join_stream_close(self.js) } }
}

#[allow(unused)]
fn _types(_: ArrayRef) {}
