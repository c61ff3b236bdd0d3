// abi_guard.h — no C++ exception may cross the C ABI (a C caller would terminate, a Rust caller is undefined
// behaviour): entry points that allocate, start threads or build strings run their body through csvsimd_guarded.
#pragma once
#include <exception>
#include <new>

#include "csvsimd.h"

// sets the thread's csvsimd_last_error() text (capi.cpp); must not throw
void csvsimd_set_last_error_noexcept(const char* what) noexcept;

template <class F>
static inline int csvsimd_guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        csvsimd_set_last_error_noexcept("out of host memory");
    } catch (const std::exception& e) {
        csvsimd_set_last_error_noexcept(e.what());
    } catch (...) {
        csvsimd_set_last_error_noexcept("unknown C++ exception");
    }
    return CSVSIMD_ERR_INTERNAL;
}
