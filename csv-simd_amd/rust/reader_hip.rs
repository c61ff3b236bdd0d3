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
}

const CSVSIMD_ERR_TAPE_CAPACITY: c_int = -11;
/// The C ABI this file was written against (include/csvsimd.h): entry points changed their argument lists between
/// versions, so a library of another version is refused instead of called.
const CSVSIMD_ABI_VERSION: u32 = 4;

/// One context per thread, created on the first `read` and kept: a context owns device scratch and pinned staging
/// (hipMalloc / hipHostMalloc: milliseconds), a read of a small file costs tens of microseconds.  A context serves one
/// call at a time, so threads do not share one.
struct Ctx(*mut CsvsimdCtx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { csvsimd_ctx_destroy(self.0) }
    }
}
thread_local! {
    static CTX: Ctx = unsafe {
        assert_eq!(csvsimd_abi_version(), CSVSIMD_ABI_VERSION, "libcsvsimd_hip.so: unexpected C ABI version");
        let mut ctx: *mut CsvsimdCtx = std::ptr::null_mut();
        assert_eq!(csvsimd_ctx_create(0, &mut ctx), 0, "csvsimd: no usable HIP device");
        Ctx(ctx)
    };
}

/// Same signature and result as `reader::read`.  Panics on a GPU/runtime failure, like the
/// reference panics on its own unsupported inputs (its signature has no `Result`).
pub fn read(memmap: &Mmap) -> StructureIndex {
    CTX.with(|ctx| unsafe {
        // first guess: one structural byte per 8 bytes of input; exact retry if the file is denser
        let mut acc: Vec<usize> = Vec::with_capacity(memmap.len() / 8 + 64);
        let mut n: u64 = 0;
        let mut rc = csvsimd_stage1_index(
            ctx.0, memmap.as_ptr(), memmap.len() as u64,
            acc.as_mut_ptr() as *mut u64, acc.capacity() as u64, &mut n, std::ptr::null_mut(),
        );
        if rc == CSVSIMD_ERR_TAPE_CAPACITY {
            acc = Vec::with_capacity(n as usize);
            rc = csvsimd_stage1_index(
                ctx.0, memmap.as_ptr(), memmap.len() as u64,
                acc.as_mut_ptr() as *mut u64, acc.capacity() as u64, &mut n, std::ptr::null_mut(),
            );
        }
        assert_eq!(rc, 0, "csvsimd_stage1_index failed");
        acc.set_len(n as usize); // usize == u64 on every target the crate supports (x86_64)
        StructureIndex(cast_vec::<usize, CodeUnitPos>(acc))
    })
}
