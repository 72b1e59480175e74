"""The few HIP runtime calls the GPU tests make themselves (ctypes on libamdhip64): copies and waits enqueued on the library's own
stream or on a producer stream of the test's, exactly as a C++ caller of include/specscan.h would enqueue them — without handing
foreign streams to torch's allocators."""
import ctypes as C

H2D, D2H = 1, 2
_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipStreamDestroy.argtypes = [C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    return _hip


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed: hipError {err}")


def copy_async(dst_ptr, src_ptr, nbytes, kind, stream):
    check(hip().hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, kind, C.c_void_p(stream)), "hipMemcpyAsync")


def stream_create():
    s = C.c_void_p()
    check(hip().hipStreamCreateWithFlags(C.byref(s), 1), "hipStreamCreateWithFlags")  # hipStreamNonBlocking
    return s.value


def stream_destroy(s):
    hip().hipStreamDestroy(C.c_void_p(s))


def stream_sync(s):
    check(hip().hipStreamSynchronize(C.c_void_p(s)), "hipStreamSynchronize")


def event_create():
    e = C.c_void_p()
    check(hip().hipEventCreateWithFlags(C.byref(e), 2), "hipEventCreateWithFlags")  # hipEventDisableTiming
    return e.value


def event_destroy(e):
    hip().hipEventDestroy(C.c_void_p(e))


def stream_wait_stream(waiter, other, event):
    """`waiter` waits for everything `other` holds now."""
    check(hip().hipEventRecord(C.c_void_p(event), C.c_void_p(other)), "hipEventRecord")
    check(hip().hipStreamWaitEvent(C.c_void_p(waiter), C.c_void_p(event), 0), "hipStreamWaitEvent")
