/* ggq_pyfast.c -- a CPython binding of ONE entry point of the C ABI, ggq_dequant (include/ggq.h), for the per-layer hot loop.
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
static ggq_dequant_fn g_dequant = NULL;

static PyObject* fast_bind(PyObject* self, PyObject* arg)
{
    (void)self;
    void* p = PyLong_AsVoidPtr(arg);
    if (p == NULL && PyErr_Occurred()) return NULL;
    g_dequant = (ggq_dequant_fn)p;
    Py_RETURN_NONE;
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
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef fast_module = {PyModuleDef_HEAD_INIT, "_ggq_fast", "fast binding of ggq_dequant", -1, fast_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__ggq_fast(void) { return PyModule_Create(&fast_module); }
