"""CelebA input files of the reference without TensorFlow.

vae/data.py:93-100 writes, and :123-131 reads, `data/celeba/{train,test}_{64x64,128x128}.tfrec`: a TFRecord file
whose records are `tf.io.serialize_tensor(image)` of one float32 [size, size, 3] image in [-1, 1].

  TFRecord framing (tensorflow/core/lib/io/record_writer.cc, public format):
      uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)
      masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8   (mod 2^32), crc32c = CRC-32/ISCSI (Castagnoli)
  serialize_tensor = a serialized `TensorProto` (tensorflow/core/framework/tensor.proto):
      field 1 dtype (varint, DT_FLOAT = 1), field 2 tensor_shape { repeated field 2 dim { field 1 size (varint) } },
      field 4 tensor_content (bytes, little-endian row-major), [field 3 version_number]

Both are restated from the published formats (TensorFlow itself cannot be installed here, so there is no
TF-written file to test against: "unpinned" like the oracle); the CRC is pinned by the CRC-32C check value
0xE3069283 of b"123456789" and the codec by a round trip.  Also the 20 000-element shuffle buffer of
vae/main.py:57-61 (tf.data.Dataset.shuffle: fill a buffer, emit a uniformly chosen slot, refill it).
"""
import struct

import numpy as np

_MASK = 0xa282ead8
_TABLE = None


def _table():
    global _TABLE
    if _TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _TABLE = t
    return _TABLE


_NATIVE = None


def _native():
    """sv_crc32c of the C ABI (slicing-by-8, ~1.5 GB/s) when the library is built; the table loop below otherwise."""
    global _NATIVE
    if _NATIVE is None:
        try:
            from . import _lib
            _NATIVE = _lib.load().sv_crc32c
        except Exception:
            _NATIVE = False
    return _NATIVE


def crc32c(data):
    """CRC-32C (Castagnoli, reflected 0x1EDC6F41) of a bytes-like object."""
    f = _native()
    if f:
        b = bytes(data)
        return int(f(b, len(b)))
    return crc32c_py(data)


def crc32c_py(data):
    t = _table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK) & 0xFFFFFFFF


def read_records(path, verify_data_crc=False):
    """Yield the raw record payloads of a TFRecord file.  The 12-byte length header is always verified;
    the payload CRC only on request (pure-Python CRC of ~49 KB per image is slow)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (n,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc32c(head[:8]) != lcrc:
                raise IOError("corrupt TFRecord length in %s" % path)
            data = f.read(n)
            tail = f.read(4)
            if len(data) < n or len(tail) < 4:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify_data_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise IOError("corrupt TFRecord payload in %s" % path)
            yield data


def write_records(path, payloads):
    with open(path, "wb") as f:
        for d in payloads:
            d = bytes(d)
            head = struct.pack("<Q", len(d))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + d + struct.pack("<I", masked_crc32c(d)))


# ---------------------------------------------------------------------------- protobuf wire format (the three types used)
def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, i
        s += 7


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


DT_FLOAT = 1


def parse_tensor(payload, out_type=np.float32):
    """tf.io.parse_tensor(x, out_type=tf.float32) for the tensors serialize_tensor writes (tensor_content form)."""
    dtype, dims, content, float_val = None, [], None, []
    for num, wt, v in _fields(memoryview(payload)):
        if num == 1 and wt == 0:
            dtype = v
        elif num == 2 and wt == 2:
            for n2, w2, v2 in _fields(v):
                if n2 == 2 and w2 == 2:                      # dim
                    size = 0
                    for n3, w3, v3 in _fields(v2):
                        if n3 == 1 and w3 == 0:
                            size = v3
                    dims.append(size)
        elif num == 4 and wt == 2:
            content = bytes(v)
        elif num == 5:                                       # repeated float_val (packed or not): small tensors only
            float_val.append(bytes(v))
    if dtype != DT_FLOAT or out_type != np.float32:
        raise TypeError("expected a DT_FLOAT tensor (vae/data.py:124), got dtype %r" % dtype)
    if content is not None:
        a = np.frombuffer(content, dtype="<f4")
    else:
        a = np.frombuffer(b"".join(float_val), dtype="<f4")
    return a.reshape(dims).astype(np.float32, copy=False)


def serialize_tensor(a):
    """tf.io.serialize_tensor for a float32 array (dtype, shape, tensor_content)."""
    a = np.ascontiguousarray(a, dtype="<f4")
    shape = b"".join(b"\x12" + _enc_varint(len(d)) + d for d in (b"\x08" + _enc_varint(int(s)) for s in a.shape))
    content = a.tobytes()
    return b"\x08" + _enc_varint(DT_FLOAT) + b"\x12" + _enc_varint(len(shape)) + shape + b"\x22" + _enc_varint(len(content)) + content


def read_celeba_tfrec_array(path, size):
    """The whole file as one [N, size, size, 3] float32 array without a per-record Python loop: the records of these
    files all have the same length (one serialize_tensor of a fixed-shape float image), so after parsing the first
    record the tensor contents are a strided view of the file.  Falls back to the record loop if the lengths differ.
    Every length header is still CRC-checked (vectorised over the distinct header values)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        return np.zeros((0, size, size, 3), np.float32)
    (n0,) = struct.unpack("<Q", raw[:8].tobytes())
    rec = 12 + n0 + 4
    first = raw[12:12 + n0].tobytes()
    img0 = parse_tensor(first)
    content = img0.astype("<f4").tobytes()
    at = first.rfind(content)                              # offset of tensor_content inside the payload
    if raw.size % rec or at < 0 or img0.size != size * size * 3:
        return np.stack(list(read_celeba_tfrec(path, size)))
    recs = raw.reshape(-1, rec)
    if not (recs[:, :12] == recs[0, :12]).all() or masked_crc32c(raw[:8].tobytes()) != struct.unpack("<I", raw[8:12].tobytes())[0]:
        return np.stack(list(read_celeba_tfrec(path, size)))
    if not (recs[:, 12:12 + at] == recs[0, 12:12 + at]).all():      # same proto prefix (dtype, shape, content length)
        return np.stack(list(read_celeba_tfrec(path, size)))
    body = np.ascontiguousarray(recs[:, 12 + at:12 + at + len(content)])
    return body.view("<f4").reshape(-1, size, size, 3)


def read_celeba_tfrec(path, size, verify_data_crc=False):
    """vae/data.py:123-131: TFRecordDataset(path).map(parse) -> float32 [size, size, 3] images."""
    for rec in read_records(path, verify_data_crc):
        yield parse_tensor(rec).reshape(size, size, 3)


def write_celeba_tfrec(path, images):
    """vae/data.py:93-100 (after its JPEG decode / crop / resize): one serialize_tensor record per image."""
    write_records(path, (serialize_tensor(np.asarray(x, np.float32)) for x in images))


def shuffle_buffer(iterable, buffer_size, seed=None):
    """tf.data.Dataset.shuffle(buffer_size) (vae/main.py:57-58): keep `buffer_size` elements, emit a uniformly random
    one and replace it with the next input; drain randomly at the end."""
    rng = np.random.default_rng(seed)
    buf = []
    for x in iterable:
        if len(buf) < buffer_size:
            buf.append(x)
            continue
        j = int(rng.integers(len(buf)))
        y, buf[j] = buf[j], x
        yield y
    while buf:
        j = int(rng.integers(len(buf)))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()
