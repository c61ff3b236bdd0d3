//! Drop-in replacement for `csv_simd::reader::read` (reference src/reader.rs:150) that runs stage 1
//! on an MI355X through libcsvsimd_hip.so (include/csvsimd.h).
//!
//! SOURCE ONLY: this image has no rustc/cargo, so this file has never been compiled here.  It is
//! the ~60-line binding a maintainer of the reference crate would add;
//! `tape.rs`, `record_source.rs` and `lib.rs::create` stay untouched because the returned
//! `StructureIndex` is bit-identical to the CPU path's: `[0, off_1, off_2, ...]`, ascending
//! `usize` offsets of every unquoted ',', CR and LF.
//!
//! build.rs:  println!("cargo:rustc-link-lib=dylib=csvsimd_hip");
use crate::stage1::{CodeUnitPos, StructureIndex};
use bytemuck::allocation::cast_vec;
use memmap::Mmap;
use std::os::raw::c_int;

#[repr(C)]
pub struct CsvsimdCtx {
    _private: [u8; 0],
}

extern "C" {
    fn csvsimd_abi_version() -> u32;
    fn csvsimd_ctx_create(device: c_int, out: *mut *mut CsvsimdCtx) -> c_int;
    fn csvsimd_ctx_destroy(ctx: *mut CsvsimdCtx);
    fn csvsimd_stage1_index(
        ctx: *mut CsvsimdCtx,
        buf: *const u8,
        len: u64,
        tape: *mut u64,
        tape_cap: u64,
        tape_len: *mut u64,
        in_quote_out: *mut u32,
    ) -> c_int;
    fn csvsimd_stage1_index_batch(ctx: *mut CsvsimdCtx, items: *mut CsvsimdHostBatchItem, n_items: u32) -> c_int;
}

/// include/csvsimd.h: csvsimd_host_batch_item
#[repr(C)]
pub struct CsvsimdHostBatchItem {
    buf: *const u8,
    len: u64,
    tape: *mut u64,
    tape_cap: u64,
    tape_len: u64,     // out
    in_quote_out: u32, // out
    status: i32,       // out
}

const CSVSIMD_ERR_TAPE_CAPACITY: c_int = -11;
/// The C ABI this file was written against (include/csvsimd.h): entry points changed their argument lists between
/// versions, so a library of another version is refused instead of called.
const CSVSIMD_ABI_VERSION: u32 = 5;

/// One context per thread, created on the first `read` and kept: a context owns device scratch and pinned staging
/// (hipMalloc / hipHostMalloc: milliseconds), a read of a small file costs tens of microseconds.  A context serves one
/// call at a time, so threads do not share one.
///
/// A thread's context is destroyed when the thread ends — except on the MAIN thread: its thread-local
/// destructors run during process teardown (if Rust runs them at all there), when the HIP runtime may already be gone, and
/// csvsimd_ctx_destroy synchronises the device and frees through it.  The main thread's context is therefore leaked to the
/// operating system at exit; a program that wants it gone earlier calls `shutdown()` on that thread while it is still
/// running.
struct Ctx(*mut CsvsimdCtx, bool /* owned by a spawned thread: destroy on drop */);
impl Drop for Ctx {
    fn drop(&mut self) {
        if self.1 && !self.0.is_null() {
            unsafe { csvsimd_ctx_destroy(self.0) }
        }
    }
}
fn on_main_thread() -> bool {
    // (the Rust runtime names the main thread "main"; no extra dependency needed for the test)
    std::thread::current().name() == Some("main")
}
/// Destroys the calling thread's context now (it is created again by the next `read`).  Safe at any point where no other
/// call of this module is running on this thread; the only way the main thread's context is ever destroyed.
pub fn shutdown() {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if let Some(ctx) = c.take() {
            unsafe { csvsimd_ctx_destroy(ctx.0) };
            std::mem::forget(ctx);
        }
    })
}
thread_local! {
    static CTX: std::cell::RefCell<Option<Ctx>> = std::cell::RefCell::new(None);
}
fn with_ctx<R>(f: impl FnOnce(*mut CsvsimdCtx) -> R) -> R {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.is_none() {
            *c = Some(unsafe { new_ctx() });
        }
        f(c.as_ref().unwrap().0)
    })
}
unsafe fn new_ctx() -> Ctx {
    {
        assert_eq!(csvsimd_abi_version(), CSVSIMD_ABI_VERSION, "libcsvsimd_hip.so: unexpected C ABI version");
        let mut ctx: *mut CsvsimdCtx = std::ptr::null_mut();
        assert_eq!(csvsimd_ctx_create(0, &mut ctx), 0, "csvsimd: no usable HIP device");
        Ctx(ctx, !on_main_thread())
    }
}

/// Same signature and result as `reader::read`.  Panics on a GPU/runtime failure, like the
/// reference panics on its own unsupported inputs (its signature has no `Result`).
pub fn read(memmap: &Mmap) -> StructureIndex {
    with_ctx(|ctx| unsafe {
        // first guess: one structural byte per 8 bytes of input; exact retry if the file is denser
        let mut acc: Vec<usize> = Vec::with_capacity(memmap.len() / 8 + 64);
        let mut n: u64 = 0;
        let mut rc = csvsimd_stage1_index(
            ctx, memmap.as_ptr(), memmap.len() as u64,
            acc.as_mut_ptr() as *mut u64, acc.capacity() as u64, &mut n, std::ptr::null_mut(),
        );
        if rc == CSVSIMD_ERR_TAPE_CAPACITY {
            acc = Vec::with_capacity(n as usize);
            rc = csvsimd_stage1_index(
                ctx, memmap.as_ptr(), memmap.len() as u64,
                acc.as_mut_ptr() as *mut u64, acc.capacity() as u64, &mut n, std::ptr::null_mut(),
            );
        }
        assert_eq!(rc, 0, "csvsimd_stage1_index failed");
        acc.set_len(n as usize); // usize == u64 on every target the crate supports (x86_64)
        StructureIndex(cast_vec::<usize, CodeUnitPos>(acc))
    })
}

/// `read` for many files at once (csvsimd_stage1_index_batch): one result per mapping, in order — each bit-identical to
/// `read(&maps[i])`.  The reference works per file (`csv_simd::create`, src/lib.rs:61-74) and its own inputs are a few
/// hundred bytes: called one by one each pays a kernel launch and a wait (30 us against 1.3 us on one CPU core); here
/// thousands of small files share one pipeline (packed groups, one copy and one batched launch per group).
pub fn read_many(maps: &[&Mmap]) -> Vec<StructureIndex> {
    with_ctx(|ctx| unsafe {
        let mut accs: Vec<Vec<usize>> = maps.iter().map(|m| Vec::with_capacity(m.len() / 4 + 32)).collect();
        let mut items: Vec<CsvsimdHostBatchItem> = maps
            .iter()
            .zip(accs.iter_mut())
            .map(|(m, a)| CsvsimdHostBatchItem {
                buf: m.as_ptr(), len: m.len() as u64, tape: a.as_mut_ptr() as *mut u64, tape_cap: a.capacity() as u64,
                tape_len: 0, in_quote_out: 0, status: 0,
            })
            .collect();
        let mut rc = csvsimd_stage1_index_batch(ctx, items.as_mut_ptr(), items.len() as u32);
        if rc == CSVSIMD_ERR_TAPE_CAPACITY {
            // the files denser than an entry per 4 bytes, again, with the exact capacity the first pass reported
            let redo: Vec<usize> = (0..items.len()).filter(|&i| items[i].status == CSVSIMD_ERR_TAPE_CAPACITY).collect();
            let mut again: Vec<CsvsimdHostBatchItem> = Vec::with_capacity(redo.len());
            for &i in &redo {
                accs[i] = Vec::with_capacity(items[i].tape_len as usize);
                again.push(CsvsimdHostBatchItem {
                    buf: items[i].buf, len: items[i].len, tape: accs[i].as_mut_ptr() as *mut u64,
                    tape_cap: accs[i].capacity() as u64, tape_len: 0, in_quote_out: 0, status: 0,
                });
            }
            rc = csvsimd_stage1_index_batch(ctx, again.as_mut_ptr(), again.len() as u32);
            for (k, &i) in redo.iter().enumerate() {
                items[i].tape_len = again[k].tape_len;
                items[i].status = again[k].status;
            }
        }
        assert_eq!(rc, 0, "csvsimd_stage1_index_batch failed");
        accs.into_iter()
            .zip(items.iter())
            .map(|(mut a, it)| {
                assert_eq!(it.status, 0);
                a.set_len(it.tape_len as usize);
                StructureIndex(cast_vec::<usize, CodeUnitPos>(a))
            })
            .collect()
    })
}
