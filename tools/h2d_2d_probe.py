"""How fast does a TIME slice of a host [B, T, V] float32 batch reach the device? (GPU box.) One contiguous hipMemcpy of the
whole batch against hipMemcpy2D of [B rows x (C x V x 4) bytes] slices with the batch's pitch, pageable and page-locked
(hipHostRegister) -- what a time-chunked ingest pipeline (copy of chunk k+1 under the decode of chunk k) could count on."""
import ctypes as C
import time

import numpy as np
import torch  # (brings the HIP runtime the process uses)

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
H2D = 1


def main():
    B, T, V = 512, 1000, 1024
    x = np.random.default_rng(0).standard_normal((B, T, V)).astype(np.float32)
    dev = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    nbytes = x.nbytes

    def whole():
        t0 = time.perf_counter()
        assert hip.hipMemcpy(dev.data_ptr(), x.ctypes.data, nbytes, H2D) == 0
        return time.perf_counter() - t0

    def sliced(chunks):
        Cn = T // chunks
        t0 = time.perf_counter()
        for k in range(chunks):
            # dst packed [B, Cn, V] at offset k * B * Cn * V; src rows of the batch, pitch T * V * 4
            rc = hip.hipMemcpy2D(dev.data_ptr() + k * B * Cn * V * 4, Cn * V * 4, x.ctypes.data + k * Cn * V * 4, T * V * 4,
                                 Cn * V * 4, B, H2D)
            assert rc == 0, rc
        return time.perf_counter() - t0

    for name, fn in (("one hipMemcpy of the batch", whole), ("hipMemcpy2D, 4 time slices", lambda: sliced(4)),
                     ("hipMemcpy2D, 8 time slices", lambda: sliced(8)), ("hipMemcpy2D, 20 time slices", lambda: sliced(20))):
        fn()
        ts = [fn() for _ in range(3)]
        print("pageable    %-32s %.1f ms  %.1f GB/s" % (name, 1e3 * min(ts), nbytes / min(ts) / 1e9), flush=True)
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(x.ctypes.data, nbytes, 0)
    print("hipHostRegister of %.1f GB: rc %d, %.1f ms" % (nbytes / 1e9, rc, 1e3 * (time.perf_counter() - t0)))
    if rc == 0:
        for name, fn in (("one hipMemcpy of the batch", whole), ("hipMemcpy2D, 8 time slices", lambda: sliced(8))):
            fn()
            ts = [fn() for _ in range(3)]
            print("page-locked %-32s %.1f ms  %.1f GB/s" % (name, 1e3 * min(ts), nbytes / min(ts) / 1e9), flush=True)
        hip.hipHostUnregister(x.ctypes.data)


if __name__ == "__main__":
    main()
