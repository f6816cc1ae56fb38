/* ggq_pyfast.c -- a CPython binding of the per-layer entry points of the C ABI (include/ggq.h): ggq_dequant, and the two opt-in fused
 * linears ggq_linear_small / ggq_linear_mfma, for the per-layer hot loop.
 *
 * ComfyUI calls dequantize_tensor once per quantized layer per forward (reference ops.py:177) and the kernel behind it runs for a
 * few microseconds, so the host side of the call is part of the hot path.  ctypes spends ~0.5 us converting the seven arguments;
 * this module takes them with METH_FASTCALL and calls straight through a function pointer.  It links against nothing: the host
 * side hands it the ADDRESS of ggq_dequant taken from the already loaded libggq_hip.so (`bind`), so there is still exactly one
 * copy of the library in the process.  Optional: when it is not built, dequant.py keeps using ctypes (same function, same result).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*ggq_dequant_fn)(int, const void*, uint64_t, void*, int, int, void*);
typedef int (*ggq_linear_small_fn)(int, const void*, uint32_t, uint32_t, const void*, uint32_t, const void*, void*, int, void*);
typedef int (*ggq_linear_mfma_fn)(int, const void*, uint32_t, uint32_t, const void*, uint32_t, const void*, void*, int, int, void*);
static ggq_dequant_fn g_dequant = NULL;
static ggq_linear_small_fn g_small = NULL;
static ggq_linear_mfma_fn g_mfma = NULL;
typedef int (*ggq_linear_mfma_ws_fn)(int, const void*, uint32_t, uint32_t, const void*, uint32_t, const void*, void*, int, int, void*, uint64_t, void*);
static ggq_linear_mfma_ws_fn g_mfma_ws = NULL;

static PyObject* fast_bind(PyObject* self, PyObject* arg)
{
    (void)self;
    void* p = PyLong_AsVoidPtr(arg);
    if (p == NULL && PyErr_Occurred()) return NULL;
    g_dequant = (ggq_dequant_fn)p;
    Py_RETURN_NONE;
}

/* bind_linear(address of ggq_linear_small, address of ggq_linear_mfma) */
static PyObject* fast_bind_linear(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 3) {
        PyErr_SetString(PyExc_TypeError, "bind_linear(addr_small, addr_mfma, addr_mfma_ws)");
        return NULL;
    }
    void* a = PyLong_AsVoidPtr(args[0]);
    void* b = PyLong_AsVoidPtr(args[1]);
    void* c = PyLong_AsVoidPtr(args[2]);
    if (PyErr_Occurred()) return NULL;
    g_small = (ggq_linear_small_fn)a;
    g_mfma = (ggq_linear_mfma_fn)b;
    g_mfma_ws = (ggq_linear_mfma_ws_fn)c;
    Py_RETURN_NONE;
}

/* integers and addresses alike (user-space addresses fit a signed 64-bit); None = NULL; a negative int (an invalid tile size, say) passes
 * through to the C entry point, which answers with GGQ_ERR_ARG as it does for ctypes */
static int as_u64s(PyObject* const* args, Py_ssize_t n, unsigned long long* out)
{
    for (Py_ssize_t i = 0; i < n; i++) {
        const long long v = (args[i] == Py_None) ? 0ll : PyLong_AsLongLong(args[i]);
        if (v == -1 && PyErr_Occurred()) return -1;
        out[i] = (unsigned long long)v;
    }
    return 0;
}

/* linear_small(qtype, packed, rows, cols, x, m, bias | None, y, dtype, stream) -> ggq_status */
static PyObject* fast_linear_small(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    unsigned long long v[10];
    if (nargs != 10 || g_small == NULL) {
        PyErr_SetString(PyExc_TypeError, "linear_small(qtype, packed, rows, cols, x, m, bias, y, dtype, stream) after bind_linear()");
        return NULL;
    }
    if (as_u64s(args, 10, v)) return NULL;
    return PyLong_FromLong(g_small((int)v[0], (const void*)(uintptr_t)v[1], (uint32_t)v[2], (uint32_t)v[3], (const void*)(uintptr_t)v[4], (uint32_t)v[5],
                                   (const void*)(uintptr_t)v[6], (void*)(uintptr_t)v[7], (int)v[8], (void*)(uintptr_t)v[9]));
}

/* linear_mfma(qtype, packed, rows, cols, x, m, bias | None, y, dtype, tile_rows, stream) -> ggq_status */
static PyObject* fast_linear_mfma(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    unsigned long long v[11];
    if (nargs != 11 || g_mfma == NULL) {
        PyErr_SetString(PyExc_TypeError, "linear_mfma(qtype, packed, rows, cols, x, m, bias, y, dtype, tile_rows, stream) after bind_linear()");
        return NULL;
    }
    if (as_u64s(args, 11, v)) return NULL;
    return PyLong_FromLong(g_mfma((int)v[0], (const void*)(uintptr_t)v[1], (uint32_t)v[2], (uint32_t)v[3], (const void*)(uintptr_t)v[4], (uint32_t)v[5],
                                  (const void*)(uintptr_t)v[6], (void*)(uintptr_t)v[7], (int)v[8], (int)v[9], (void*)(uintptr_t)v[10]));
}

/* linear_mfma_ws(qtype, packed, rows, cols, x, m, bias | None, y, dtype, tile_rows, workspace | None, workspace_bytes, stream) -> ggq_status */
static PyObject* fast_linear_mfma_ws(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    unsigned long long v[13];
    if (nargs != 13 || g_mfma_ws == NULL) {
        PyErr_SetString(PyExc_TypeError, "linear_mfma_ws(qtype, packed, rows, cols, x, m, bias, y, dtype, tile_rows, workspace, workspace_bytes, stream) after bind_linear()");
        return NULL;
    }
    if (as_u64s(args, 13, v)) return NULL;
    return PyLong_FromLong(g_mfma_ws((int)v[0], (const void*)(uintptr_t)v[1], (uint32_t)v[2], (uint32_t)v[3], (const void*)(uintptr_t)v[4], (uint32_t)v[5],
                                     (const void*)(uintptr_t)v[6], (void*)(uintptr_t)v[7], (int)v[8], (int)v[9], (void*)(uintptr_t)v[10], (uint64_t)v[11],
                                     (void*)(uintptr_t)v[12]));
}

/* dequant(qtype, packed_ptr, n_blocks, out_ptr, compute_dtype, out_dtype, stream) -> ggq_status */
static PyObject* fast_dequant(PyObject* self, PyObject* const* args, Py_ssize_t nargs)
{
    (void)self;
    if (nargs != 7) {
        PyErr_SetString(PyExc_TypeError, "dequant(qtype, packed, n_blocks, out, compute_dtype, out_dtype, stream)");
        return NULL;
    }
    if (g_dequant == NULL) {
        PyErr_SetString(PyExc_RuntimeError, "_ggq_fast: bind() has not been called");
        return NULL;
    }
    const long qtype = PyLong_AsLong(args[0]);
    const unsigned long long packed = PyLong_AsUnsignedLongLong(args[1]);
    const unsigned long long n_blocks = PyLong_AsUnsignedLongLong(args[2]);
    const unsigned long long out = PyLong_AsUnsignedLongLong(args[3]);
    const long cd = PyLong_AsLong(args[4]);
    const long od = PyLong_AsLong(args[5]);
    const unsigned long long stream = PyLong_AsUnsignedLongLong(args[6]);
    if (PyErr_Occurred()) return NULL;
    const int rc = g_dequant((int)qtype, (const void*)(uintptr_t)packed, (uint64_t)n_blocks, (void*)(uintptr_t)out, (int)cd, (int)od, (void*)(uintptr_t)stream);
    return PyLong_FromLong(rc);
}

static PyMethodDef fast_methods[] = {
    {"bind", (PyCFunction)fast_bind, METH_O, "bind(address of ggq_dequant)"},
    {"dequant", (PyCFunction)(void (*)(void))fast_dequant, METH_FASTCALL, "ggq_dequant through a plain function pointer"},
    {"bind_linear", (PyCFunction)(void (*)(void))fast_bind_linear, METH_FASTCALL, "bind_linear(address of ggq_linear_small, address of ggq_linear_mfma, address of ggq_linear_mfma_ws)"},
    {"linear_small", (PyCFunction)(void (*)(void))fast_linear_small, METH_FASTCALL, "ggq_linear_small through a plain function pointer"},
    {"linear_mfma", (PyCFunction)(void (*)(void))fast_linear_mfma, METH_FASTCALL, "ggq_linear_mfma through a plain function pointer"},
    {"linear_mfma_ws", (PyCFunction)(void (*)(void))fast_linear_mfma_ws, METH_FASTCALL, "ggq_linear_mfma_ws through a plain function pointer"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_ggq_fast", "fast binding of ggq_dequant", -1, fast_methods, NULL, NULL, NULL, NULL};

/* ABI: the version of include/ggq.h whose three signatures this file was written against.  _native.fast() refuses a binary whose
 * constant differs from the loaded library's ggq_abi_version(): a stale _ggq_fast would call through raw pointers with the wrong
 * argument lists. */
#define GGQ_FAST_ABI 11

PyMODINIT_FUNC PyInit__ggq_fast(void)
{
    PyObject* m = PyModule_Create(&fast_module);
    if (m != NULL && PyModule_AddIntConstant(m, "ABI", GGQ_FAST_ABI) < 0) {
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
