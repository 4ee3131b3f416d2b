/* _so3fast -- a thin CPython entry for the enqueue-only C-ABI calls of libso3proj.so.
 *
 * The reference's batch sizes are 64-512 (Iterative/main.py:216, UPNA/main.py:126, 3D-Pose/configs/example.yaml:3): a call of
 * the library then costs 3-4 us of GPU time, and ctypes' per-call argument conversion (~2.5 us for nine arguments) is on the
 * critical path of every training step.  call(address, a0, ..., a11) invokes the function at `address` with up to twelve
 * integer-class arguments (device pointers, sizes, the stream; None is a null pointer) and returns its int status -- METH_FASTCALL,
 * no keyword parsing, no GIL release (the calls only enqueue).  Functions with floating-point parameters go through ctypes.
 *
 * x86-64 System V: integer-class arguments beyond a callee's own are ignored (six in registers, the rest in caller-cleaned stack
 * slots), so one twelve-argument call type serves every entry point.  Built by poseestimation_amd/build.py with gcc.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*so3_fn_t)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t,
                        intptr_t);

/* The one thing the sanitizers report in this file (round 6: tools/sanitize_cpu.py, -fsanitize=function) is its premise: the callee is
 * invoked through the twelve-integer type above, not through its own prototype.  ISO C leaves that undefined; the x86-64 System V calling
 * convention defines it (see the header of this file), and libffi / ctypes do the same one level down.  The check is switched off for
 * this function alone -- everything else in it (argument conversion, the error paths) runs instrumented. */
#if defined(__clang__)
__attribute__((no_sanitize("function")))
#endif
static PyObject *so3fast_call(PyObject *self, PyObject *const *args, Py_ssize_t nargs) {
    intptr_t a[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)self;
    if (nargs < 1 || nargs > 13) {
        PyErr_SetString(PyExc_TypeError, "call(address, up to 12 integer arguments)");
        return NULL;
    }
    const uintptr_t addr = (uintptr_t)PyLong_AsUnsignedLongLongMask(args[0]);
    if (addr == 0 || PyErr_Occurred()) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "null function address");
        return NULL;
    }
    for (Py_ssize_t i = 1; i < nargs; ++i) {
        PyObject *o = args[i];
        if (o == Py_None) continue;
        const unsigned long long v = PyLong_AsUnsignedLongLongMask(o);      /* two's complement for negative ints */
        if (v == (unsigned long long)-1 && PyErr_Occurred()) return NULL;
        a[i - 1] = (intptr_t)v;
    }
    const int rc = ((so3_fn_t)addr)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]);
    return PyLong_FromLong(rc);
}

static PyMethodDef so3fast_methods[] = {
    {"call", (PyCFunction)(void (*)(void))so3fast_call, METH_FASTCALL, "call(address, *integer_args) -> int status"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef so3fast_module = {PyModuleDef_HEAD_INIT, "_so3fast", "fast enqueue calls into libso3proj.so", -1, so3fast_methods,
                                            NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__so3fast(void) { return PyModule_Create(&so3fast_module); }
