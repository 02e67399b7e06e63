"""Keras HDF5 weight files (model.save_weights('models/<run>.h5'), vae/trainer.py:421) without h5py.

h5py / TensorFlow are not installable here, but the HDF5 C library itself ships with the image
(/opt/conda/lib/libhdf5.so, 1.10.x); this module binds the dozen calls the Keras weight layout needs through
ctypes.  Layout written/read (tensorflow 2.0 keras/saving/hdf5_format.py, save_weights_to_hdf5_group /
load_weights_from_hdf5_group -- third-party, restated from its published format):

    /                       attrs: layer_names = [b'encoder', ...] (fixed-length strings), backend = b'tensorflow',
                                   keras_version = b'2.2.4-tf'
    /<layer>/               attrs: weight_names = [b'<weight name>', ...]
    /<layer>/<weight name>  float32 dataset in the Keras layout (conv HWIO, dense [in,out]); the weight name contains
                            '/' so it nests groups, e.g. /encoder/lg_vae/encoder/conv2d/kernel:0

Loading goes by ORDER (layer_names, then each layer's weight_names), exactly as Keras does, so the auto-generated
layer uids in the names (conv2d_3, dense_2 ...: [TF-2.0 semantics], not verifiable here) never matter for a round trip
and a file written by the reference itself loads the same way.  If libhdf5 is absent `available()` is False and the
model falls back to its .npz container.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_lib = None
_ids = {}
hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 2, 0, 0, 0


def _find():
    cands = []
    env = os.environ.get("SV_LIBHDF5")
    if env:
        cands.append(env)
    cands += sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for c in cands:
        try:
            return C.CDLL(c)
        except OSError:
            continue
    return None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = _find()
    if lib is None:
        raise OSError("libhdf5 not found (set SV_LIBHDF5=/path/to/libhdf5.so); the .npz container needs no library")
    sig = {
        "H5open": (C.c_int, []), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fclose": (C.c_int, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (C.c_int, [hid_t]), "H5Screate": (hid_t, [C.c_int]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dget_space": (hid_t, [hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dclose": (C.c_int, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aclose": (C.c_int, [hid_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tclose": (C.c_int, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (C.c_int, [hid_t, C.c_uint]), "H5Pclose": (C.c_int, [hid_t]),
        "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise OSError("H5open failed")
    lib.H5Eset_auto2(0, None, None)                     # errors come back as negative ids; no stderr stack dumps
    for g in ("H5T_NATIVE_FLOAT_g", "H5T_IEEE_F32LE_g", "H5T_C_S1_g", "H5P_CLS_LINK_CREATE_ID_g"):
        _ids[g] = hid_t.in_dll(lib, g).value
    _lib = lib
    return lib


def available():
    try:
        _load()
        return True
    except OSError:
        return False


def _ck(v, what):
    if v < 0:
        raise IOError("HDF5: %s failed" % what)
    return v


def _write_str_attr(lib, obj, name, values, scalar=False):
    """h5py semantics of attrs[name] = np.array([b'..', ...]) / = b'..': fixed-length, null-padded strings."""
    vals = [values] if scalar else list(values)
    width = max([len(v) for v in vals] + [1])
    t = _ck(lib.H5Tcopy(_ids["H5T_C_S1_g"]), "H5Tcopy")
    _ck(lib.H5Tset_size(t, width), "H5Tset_size")
    if scalar:
        sp = _ck(lib.H5Screate(H5S_SCALAR), "H5Screate")
    else:
        dims = (hsize_t * 1)(len(vals))
        sp = _ck(lib.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _ck(lib.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2 " + name)
    buf = b"".join(v.ljust(width, b"\0") for v in vals) or b"\0"
    _ck(lib.H5Awrite(a, t, C.c_char_p(buf)), "H5Awrite " + name)
    lib.H5Aclose(a); lib.H5Sclose(sp); lib.H5Tclose(t)


def _read_str_attr(lib, obj, name):
    a = _ck(lib.H5Aopen(obj, name.encode(), H5P_DEFAULT), "H5Aopen " + name)
    t = lib.H5Aget_type(a)
    sp = lib.H5Aget_space(a)
    n = int(lib.H5Sget_simple_extent_npoints(sp))
    if lib.H5Tis_variable_str(t) > 0:                   # newer h5py writes str attributes as variable-length
        ptrs = (C.c_char_p * n)()
        _ck(lib.H5Aread(a, t, ptrs), "H5Aread " + name)
        out = [bytes(p) if p is not None else b"" for p in ptrs]
    else:
        width = int(lib.H5Tget_size(t))
        buf = C.create_string_buffer(max(n * width, 1))
        _ck(lib.H5Aread(a, t, buf), "H5Aread " + name)
        out = [buf.raw[i * width:(i + 1) * width].split(b"\0")[0] for i in range(n)]
    lib.H5Sclose(sp); lib.H5Tclose(t); lib.H5Aclose(a)
    return out


def save_keras_weights(path, layers, backend=b"tensorflow", keras_version=b"2.2.4-tf"):
    """layers: [(layer_name, [(weight_name, float32 array), ...]), ...] in Keras order."""
    lib = _load()
    f = _ck(lib.H5Fcreate(str(path).encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate " + str(path))
    try:
        lcpl = _ck(lib.H5Pcreate(_ids["H5P_CLS_LINK_CREATE_ID_g"]), "H5Pcreate")
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        _write_str_attr(lib, f, "layer_names", [n.encode("utf8") for n, _ in layers])
        _write_str_attr(lib, f, "backend", backend, scalar=True)
        _write_str_attr(lib, f, "keras_version", keras_version, scalar=True)
        for lname, weights in layers:
            g = _ck(lib.H5Gcreate2(f, lname.encode("utf8"), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2 " + lname)
            _write_str_attr(lib, g, "weight_names", [w.encode("utf8") for w, _ in weights])
            for wname, val in weights:
                val = np.ascontiguousarray(val, dtype="<f4")
                dims = (hsize_t * max(val.ndim, 1))(*val.shape)
                sp = _ck(lib.H5Screate_simple(val.ndim, dims, None) if val.ndim else lib.H5Screate(H5S_SCALAR), "H5Screate_simple")
                d = _ck(lib.H5Dcreate2(g, wname.encode("utf8"), _ids["H5T_IEEE_F32LE_g"], sp, lcpl, H5P_DEFAULT, H5P_DEFAULT),
                        "H5Dcreate2 " + wname)
                _ck(lib.H5Dwrite(d, _ids["H5T_NATIVE_FLOAT_g"], H5S_ALL, H5S_ALL, H5P_DEFAULT, val.ctypes.data_as(C.c_void_p)),
                    "H5Dwrite " + wname)
                lib.H5Dclose(d); lib.H5Sclose(sp)
            lib.H5Gclose(g)
        lib.H5Pclose(lcpl)
    finally:
        lib.H5Fclose(f)
    return str(path)


def load_keras_weights(path):
    """-> [(layer_name, [(weight_name, float32 array), ...]), ...] in file order (layer_names / weight_names attributes)."""
    lib = _load()
    f = _ck(lib.H5Fopen(str(path).encode(), H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen " + str(path))
    out = []
    try:
        root = f
        if lib.H5Aexists(f, b"layer_names") <= 0:       # model.save() files keep the weights under /model_weights
            root = _ck(lib.H5Gopen2(f, b"model_weights", H5P_DEFAULT), "no layer_names attribute and no /model_weights group")
        for lname in _read_str_attr(lib, root, "layer_names"):
            g = _ck(lib.H5Gopen2(root, lname, H5P_DEFAULT), "H5Gopen2 " + lname.decode())
            ws = []
            for wname in _read_str_attr(lib, g, "weight_names"):
                d = _ck(lib.H5Dopen2(g, wname, H5P_DEFAULT), "H5Dopen2 " + wname.decode())
                sp = lib.H5Dget_space(d)
                nd = lib.H5Sget_simple_extent_ndims(sp)
                dims = (hsize_t * max(nd, 1))()
                if nd:
                    lib.H5Sget_simple_extent_dims(sp, dims, None)
                arr = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=np.float32)
                _ck(lib.H5Dread(d, _ids["H5T_NATIVE_FLOAT_g"], H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)),
                    "H5Dread " + wname.decode())
                lib.H5Sclose(sp); lib.H5Dclose(d)
                ws.append((wname.decode("utf8"), arr))
            lib.H5Gclose(g)
            out.append((lname.decode("utf8"), ws))
        if root != f:
            lib.H5Gclose(root)
    finally:
        lib.H5Fclose(f)
    return out
