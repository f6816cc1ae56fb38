"""The GGUF container parser (csrc/ggq_gguf.hip, include/ggq_gguf.h) against the PUBLISHED container format, independently of
tests/gguf_writer.py: every file below is written out byte by byte, in hex, each field commented with the rule of the GGUF
specification (ggml docs/gguf.md) it follows.  Nothing here calls the writer, so reader and writer cannot share a misreading
(Next #3).  What stays unpinned is only agreement with the gguf-py *implementation* itself (absent from this image
and from /root/reference): where gguf-py is known to be more lenient than the specification, the case says so.

The rules used (spec section "File structure"):
  * magic 'G' 'G' 'U' 'F' = 47 47 55 46, then version u32, tensor_count u64, metadata_kv_count u64 -- little-endian; v1 had u32 counts,
    v2 widened them to u64, v3 added big-endian files (same layout, every number byte-swapped);
  * string = u64 length + bytes, no terminator; keys are strings;
  * value types UINT8 0, INT8 1, UINT16 2, INT16 3, UINT32 4, INT32 5, FLOAT32 6, BOOL 7 (one byte), STRING 8, ARRAY 9 (u32 element
    type, u64 count, elements), UINT64 10, INT64 11, FLOAT64 12;
  * tensor info = name string, n_dimensions u32, dimensions u64[n], ggml type u32, offset u64 -- the offset is relative to the START OF
    THE TENSOR-DATA SECTION and a multiple of the alignment;
  * `general.alignment` (u32, a multiple of 8; 32 when absent); the data section starts at the first multiple of it after the infos.
"""
import pytest
import torch


def H(text):
    """hex with '#' comments -> bytes"""
    return bytes.fromhex("".join(line.split("#")[0] for line in text.splitlines()))


@pytest.fixture()
def gf(pkg):
    return pkg.gguf_file


def _open(gf, tmp_path, data, name="f.gguf"):
    p = tmp_path / name
    p.write_bytes(data)
    return gf.GGUFFile(str(p))


def _rejected(gf, tmp_path, data):
    with pytest.raises(ValueError, match="GGUF|format|truncated|inconsistent"):
        _open(gf, tmp_path, data, "bad.gguf").close()


# ------------------------------------------------------------------------------------------------ 1. every scalar type, v3
SCALARS = H("""
47 47 55 46                                   # magic
03 00 00 00                                   # version 3
00 00 00 00 00 00 00 00                       # tensor_count 0
0D 00 00 00 00 00 00 00                       # metadata_kv_count 13                                  -> offset 24
02 00 00 00 00 00 00 00  75 38                # key "u8"
00 00 00 00                                   # type 0 UINT8
C8                                            # 200                                                  -> 39
02 00 00 00 00 00 00 00  69 38                # key "i8"
01 00 00 00                                   # type 1 INT8
F9                                            # -7                                                   -> 54
03 00 00 00 00 00 00 00  75 31 36             # key "u16"
02 00 00 00                                   # type 2 UINT16
FF FF                                         # 65535                                                -> 71
03 00 00 00 00 00 00 00  69 31 36             # key "i16"
03 00 00 00                                   # type 3 INT16
FD FF                                         # -3                                                   -> 88
03 00 00 00 00 00 00 00  75 33 32             # key "u32"
04 00 00 00                                   # type 4 UINT32
00 28 6B EE                                   # 4 000 000 000 = 0xEE6B2800                           -> 107
03 00 00 00 00 00 00 00  69 33 32             # key "i32"
05 00 00 00                                   # type 5 INT32
FB FF FF FF                                   # -5                                                   -> 126
03 00 00 00 00 00 00 00  66 33 32             # key "f32"
06 00 00 00                                   # type 6 FLOAT32
00 00 C0 3F                                   # 1.5 = 0x3FC00000                                     -> 145
01 00 00 00 00 00 00 00  62                   # key "b"
07 00 00 00                                   # type 7 BOOL: one byte
01                                            # true                                                 -> 159
01 00 00 00 00 00 00 00  73                   # key "s"
08 00 00 00                                   # type 8 STRING
04 00 00 00 00 00 00 00  66 6C 75 78          # "flux"                                               -> 184
03 00 00 00 00 00 00 00  75 36 34             # key "u64"
0A 00 00 00                                   # type 10 UINT64
05 00 00 00 00 00 00 80                       # 2^63 + 5                                             -> 207
03 00 00 00 00 00 00 00  69 36 34             # key "i64"
0B 00 00 00                                   # type 11 INT64
FE FF FF FF FF FF FF FF                       # -2                                                   -> 230
03 00 00 00 00 00 00 00  66 36 34             # key "f64"
0C 00 00 00                                   # type 12 FLOAT64
00 00 00 00 00 00 E0 BF                       # -0.5 = 0xBFE0000000000000                            -> 253
01 00 00 00 00 00 00 00  61                   # key "a"
09 00 00 00                                   # type 9 ARRAY
05 00 00 00                                   #   element type 5 INT32
02 00 00 00 00 00 00 00                       #   count 2 (elements, not bytes)
00 0C 00 00  FF FF FF FF                      #   3072, -1                                           -> 286
00 00                                         # padding to the alignment (32): 288
""")
SCALARS_FIELD_ENDS = [24, 39, 54, 71, 88, 107, 126, 145, 159, 184, 207, 230, 253, 286]


def test_every_scalar_type_v3(gf, tmp_path):
    assert len(SCALARS) == 288
    with _open(gf, tmp_path, SCALARS) as f:
        assert (f.version, f.alignment, f.n_kv, len(f.tensors), f.data_offset, f.data_bytes) == (3, 32, 13, 0, 288, 0)
        assert f.keys() == ["u8", "i8", "u16", "i16", "u32", "i32", "f32", "b", "s", "u64", "i64", "f64", "a"]
        want = {"u8": (gf.UINT8, 200), "i8": (gf.INT8, -7), "u16": (gf.UINT16, 65535), "i16": (gf.INT16, -3), "u32": (gf.UINT32, 4000000000),
                "i32": (gf.INT32, -5), "f32": (gf.FLOAT32, 1.5), "b": (gf.BOOL, True), "s": (gf.STRING, "flux"), "u64": (gf.UINT64, 2**63 + 5),
                "i64": (gf.INT64, -2), "f64": (gf.FLOAT64, -0.5)}
        for k, (t, v) in want.items():
            fld = f.get_field(k)
            assert fld.types == [t] and fld.value == v and type(fld.value) is type(v), k
        assert f.get_field("a") == gf.GGUFField("a", [gf.ARRAY, gf.INT32], (3072, -1))


def test_metadata_only_file_needs_no_trailing_padding(gf, tmp_path):
    with _open(gf, tmp_path, SCALARS[:286]) as f:
        assert (f.n_kv, len(f.tensors), f.data_bytes) == (13, 0, 0)


def test_truncation_inside_the_metadata(gf, tmp_path):
    """Cut at every field boundary and at every single byte in between: 13 pairs were promised, so every shorter file is an error."""
    for cut in SCALARS_FIELD_ENDS[:-1] + list(range(0, 286)):
        _rejected(gf, tmp_path, SCALARS[:cut])


# ------------------------------------------------------------------------------------------------ 2. a v2 file with one tensor
V2_ONE_TENSOR = H("""
47 47 55 46                                   # magic
02 00 00 00                                   # version 2: same layout as v3 (u64 counts and lengths)
01 00 00 00 00 00 00 00                       # tensor_count 1
01 00 00 00 00 00 00 00                       # metadata_kv_count 1                                   -> 24
14 00 00 00 00 00 00 00                       # key length 20
67 65 6E 65 72 61 6C 2E 61 72 63 68 69 74 65 63 74 75 72 65   # "general.architecture"
08 00 00 00                                   # STRING
03 00 00 00 00 00 00 00  73 64 31             # "sd1"                                                -> 67
01 00 00 00 00 00 00 00  74                   # tensor name "t"
01 00 00 00                                   # n_dimensions 1
03 00 00 00 00 00 00 00                       # dimensions[0] = 3
00 00 00 00                                   # ggml type 0 = F32
00 00 00 00 00 00 00 00                       # offset 0 (from the start of the data section)        -> 100
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00   # 28 bytes of padding -> 128 = data section
00 00 80 3F  00 00 20 C0  00 00 50 40         # 1.0, -2.5, 3.25                                      -> 140
""")


def test_v2_header_and_one_f32_tensor(pkg, gf, tmp_path):
    assert len(V2_ONE_TENSOR) == 140
    with _open(gf, tmp_path, V2_ONE_TENSOR) as f:
        assert (f.version, f.alignment, f.n_kv, f.data_offset, f.data_bytes, f.file_bytes) == (2, 32, 1, 128, 12, 140)
        assert f.get_field("general.architecture").value == "sd1"
        (t,) = f.tensors
        assert (t.name, t.tensor_type, t.shape, t.offset, t.nbytes, t.n_elements) == ("t", pkg.qtypes.Q.F32, (3,), 0, 12, 3)
        assert t.data.view(torch.float32).tolist() == [1.0, -2.5, 3.25]
    v3 = V2_ONE_TENSOR[:4] + b"\x03" + V2_ONE_TENSOR[5:]                # byte 4 = the version: v3 reads the same
    with _open(gf, tmp_path, v3) as f:
        assert f.version == 3 and f.tensors[0].data.view(torch.float32).tolist() == [1.0, -2.5, 3.25]


def test_unsupported_versions_and_magic(gf, tmp_path):
    for version in (0, 1, 4, 0x03000000):                               # v1 had u32 counts; 0x03000000 = a byte-swapped 3
        _rejected(gf, tmp_path, V2_ONE_TENSOR[:4] + version.to_bytes(4, "little") + V2_ONE_TENSOR[8:])
    _rejected(gf, tmp_path, b"GGML" + V2_ONE_TENSOR[4:])
    _rejected(gf, tmp_path, b"gguf" + V2_ONE_TENSOR[4:])


# ------------------------------------------------------------------------------------------------ 3. string arrays, nesting
STRING_ARRAY = H("""
47 47 55 46  03 00 00 00                      # magic, version 3
00 00 00 00 00 00 00 00                       # no tensors
01 00 00 00 00 00 00 00                       # one pair                                              -> 24
06 00 00 00 00 00 00 00  74 6F 6B 65 6E 73    # key "tokens"
09 00 00 00                                   # ARRAY
08 00 00 00                                   #   of STRING
03 00 00 00 00 00 00 00                       #   3 elements
05 00 00 00 00 00 00 00  3C 70 61 64 3E       #   "<pad>"
00 00 00 00 00 00 00 00                       #   "" (length 0, no bytes)
06 00 00 00 00 00 00 00  E2 96 81 74 68 65    #   U+2581 "the": the length counts BYTES of UTF-8, not characters   -> 89
""")

NESTED_ARRAY = H("""
47 47 55 46  03 00 00 00
00 00 00 00 00 00 00 00
01 00 00 00 00 00 00 00
01 00 00 00 00 00 00 00  6E                   # key "n"
09 00 00 00                                   # ARRAY
09 00 00 00                                   #   of ARRAY (the specification allows nesting; no file of this ecosystem uses it)
01 00 00 00 00 00 00 00                       #   1 element:
04 00 00 00                                   #     ARRAY of UINT32
01 00 00 00 00 00 00 00                       #     1 element
07 00 00 00                                   #     7
""")


def test_string_arrays(gf, tmp_path):
    assert len(STRING_ARRAY) == 89
    with _open(gf, tmp_path, STRING_ARRAY) as f:
        assert f.get_field("tokens") == gf.GGUFField("tokens", [gf.ARRAY, gf.STRING], ("<pad>", "", "▁the"))
    for cut in (46, 54, 62, 67, 75, 83, 88):                           # inside / between the elements
        _rejected(gf, tmp_path, STRING_ARRAY[:cut])


def test_nested_arrays_are_refused_not_misread(gf, tmp_path):
    """Documented limit (include/ggq_gguf.h): an array of arrays is answered with GGQ_ERR_FORMAT -- never parsed as something else."""
    _rejected(gf, tmp_path, NESTED_ARRAY)


def test_unknown_value_type_is_refused(gf, tmp_path):
    _rejected(gf, tmp_path, SCALARS[:34] + bytes([13]) + SCALARS[35:])    # byte 34 = type of the first pair: 13 is not a value type


# ------------------------------------------------------------------------------------------------ 4. general.alignment != 32
ALIGN64 = H("""
47 47 55 46  03 00 00 00                      # magic, version 3
02 00 00 00 00 00 00 00                       # 2 tensors
01 00 00 00 00 00 00 00                       # 1 pair                                                -> 24
11 00 00 00 00 00 00 00                       # key length 17
67 65 6E 65 72 61 6C 2E 61 6C 69 67 6E 6D 65 6E 74   # "general.alignment"                            -> 49
04 00 00 00                                   # UINT32                                                -> 53
40 00 00 00                                   # 64                                                    -> 57
08 00 00 00 00 00 00 00  61 2E 77 65 69 67 68 74   # name "a.weight"                                  -> 73
01 00 00 00                                   # 1 dimension                                           -> 77
02 00 00 00 00 00 00 00                       # [2]                                                   -> 85
00 00 00 00                                   # F32                                                   -> 89
00 00 00 00 00 00 00 00                       # offset 0                                              -> 97
08 00 00 00 00 00 00 00  62 2E 77 65 69 67 68 74   # name "b.weight"                                  -> 113
01 00 00 00                                   # 1 dimension                                           -> 117
04 00 00 00 00 00 00 00                       # [4]                                                   -> 125
01 00 00 00                                   # ggml type 1 = F16                                     -> 129
40 00 00 00 00 00 00 00                       # offset 64: the next multiple of the alignment after a.weight's 8 bytes   -> 137
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00  # 55 bytes of padding: 137 -> 192 (the next multiple of 64; with the default 32 it would be 160)
00 00 80 3F  00 00 00 40                      # a.weight = 1.0, 2.0   at data + 0                     -> 200
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00   # 56 bytes of padding -> 256
00 3C  00 40  00 BC  00 38                    # b.weight = 1.0, 2.0, -1.0, 0.5 (fp16) at data + 64 = file offset 256   -> 264
""")
ALIGN64_FIELD_ENDS = [4, 8, 16, 24, 32, 49, 53, 57, 65, 73, 77, 85, 89, 97, 105, 113, 117, 125, 129, 137, 192, 200, 256]


def test_alignment_key_moves_the_data_section_and_offsets_are_relative_to_it(pkg, gf, tmp_path):
    assert len(ALIGN64) == 264
    Q = pkg.qtypes.Q
    with _open(gf, tmp_path, ALIGN64) as f:
        assert (f.alignment, f.data_offset, f.data_bytes) == (64, 192, 72)
        a, b = f.tensors
        assert (a.name, a.tensor_type, a.shape, a.offset, a.nbytes) == ("a.weight", Q.F32, (2,), 0, 8)
        assert (b.name, b.tensor_type, b.shape, b.offset, b.nbytes) == ("b.weight", Q.F16, (4,), 64, 8)
        assert a.data.view(torch.float32).tolist() == [1.0, 2.0]
        assert b.data.view(torch.float16).tolist() == [1.0, 2.0, -1.0, 0.5]          # read at 192 + 64, not at 64


def test_tensor_offsets_must_be_multiples_of_the_alignment(gf, tmp_path):
    # byte 129 = low byte of b.weight's offset: 32 is a multiple of the default alignment but not of this file's 64
    _rejected(gf, tmp_path, ALIGN64[:129] + b"\x20" + ALIGN64[130:])


def test_alignment_values(gf, tmp_path):
    """`general.alignment` "must be a multiple of 8" (spec); 0 and 12 are not.  24 is: the data section then starts at 144."""
    for bad in (0, 12, 4):
        _rejected(gf, tmp_path, ALIGN64[:53] + bad.to_bytes(4, "little") + ALIGN64[57:])
    ok = bytearray(ALIGN64[:137])
    ok[53:57] = (24).to_bytes(4, "little")
    ok[129:137] = (24).to_bytes(8, "little")                            # b.weight at data + 24
    ok += b"\0" * 7                                                      # 137 -> 144 = 6 * 24
    ok += H("00 00 80 3F 00 00 00 40") + b"\0" * 16 + H("00 3C 00 40 00 BC 00 38")
    with _open(gf, tmp_path, bytes(ok)) as f:
        assert (f.alignment, f.data_offset) == (24, 144)
        assert f.tensors[1].data.view(torch.float16).tolist() == [1.0, 2.0, -1.0, 0.5]
    # the key must be a UINT32 (gguf-py: "Bad type for general.alignment field"): the same value typed UINT64 is refused
    as_u64 = ALIGN64[:49] + H("0A 00 00 00  40 00 00 00 00 00 00 00") + ALIGN64[57:]
    _rejected(gf, tmp_path, as_u64)


def test_truncation_at_every_field_boundary_and_every_byte(gf, tmp_path):
    """Header, pair, both tensor infos, the padding, the data of either tensor: a file cut anywhere is an error, never a fault and never
    a shorter file read as if complete."""
    for cut in ALIGN64_FIELD_ENDS + list(range(0, 264)):
        _rejected(gf, tmp_path, ALIGN64[:cut])
    _open(gf, tmp_path, ALIGN64 + b"\0" * 5).close()                     # trailing bytes after the last tensor are not an error


# ------------------------------------------------------------------------------------------------ 5. 0-dim and 4-dim tensors, block types
DIMS_HEAD = H("""
47 47 55 46  03 00 00 00
02 00 00 00 00 00 00 00                       # 2 tensors
00 00 00 00 00 00 00 00                       # no metadata                                           -> 24
06 00 00 00 00 00 00 00  73 63 61 6C 61 72    # name "scalar"
00 00 00 00                                   # n_dimensions 0: no dimension words follow; one element (empty product)
00 00 00 00                                   # F32
00 00 00 00 00 00 00 00                       # offset 0                                              -> 54
02 00 00 00 00 00 00 00  74 34                # name "t4"
04 00 00 00                                   # n_dimensions 4
02 00 00 00 00 00 00 00                       # dimensions[0] = 2: the FASTEST-varying one (ggml order)
03 00 00 00 00 00 00 00
04 00 00 00 00 00 00 00
05 00 00 00 00 00 00 00                       # dimensions[3] = 5: the slowest -- torch shape (5, 4, 3, 2)
01 00 00 00                                   # F16
20 00 00 00 00 00 00 00                       # offset 32                                             -> 112
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00   # padding -> 128
00 00 E0 40                                   # scalar = 7.0
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00   # padding -> data + 32
""")
DIMS = DIMS_HEAD + bytes(range(240))                                     # t4: 2*3*4*5 = 120 halves


def test_zero_and_four_dimensional_tensors(pkg, gf, tmp_path):
    assert len(DIMS_HEAD) == 160
    with _open(gf, tmp_path, DIMS) as f:
        s, t = f.tensors
        assert (s.name, s.shape, s.n_elements, s.nbytes) == ("scalar", (), 1, 4) and s.data.view(torch.float32).item() == 7.0
        assert (t.name, t.shape, t.n_elements, t.nbytes, t.offset) == ("t4", (2, 3, 4, 5), 120, 240, 32)
        assert bytes(t.data.numpy()) == bytes(range(240))
    _rejected(gf, tmp_path, DIMS[:-1])                                   # one byte short of t4's data
    _rejected(gf, tmp_path, DIMS[:64] + b"\x09" + DIMS[65:])             # byte 64 = n_dimensions of t4: 9 > GGQ_GGUF_MAX_DIMS


Q4K_HEAD = H("""
47 47 55 46  03 00 00 00
01 00 00 00 00 00 00 00
00 00 00 00 00 00 00 00                       #                                                       -> 24
01 00 00 00 00 00 00 00  77                   # name "w"
02 00 00 00                                   # 2 dimensions
00 01 00 00 00 00 00 00                       # dimensions[0] = 256 columns (one Q4_K super-block per row)
02 00 00 00 00 00 00 00                       # dimensions[1] = 2 rows -> torch shape (2, 256)
0C 00 00 00                                   # ggml type 12 = Q4_K: 256 elements in 144 bytes
00 00 00 00 00 00 00 00                       # offset 0                                              -> 65
00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00   # 31 bytes of padding -> 96
""")


def test_block_quantized_tensor_bytes(pkg, gf, tmp_path):
    payload = bytes((7 * i + 3) & 255 for i in range(288))               # 2 super-blocks of 144 bytes
    assert len(Q4K_HEAD) == 96
    with _open(gf, tmp_path, Q4K_HEAD + payload) as f:
        (w,) = f.tensors
        assert (w.tensor_type, w.shape, w.n_elements, w.nbytes) == (pkg.qtypes.Q.Q4_K, (256, 2), 512, 288)
        assert bytes(w.data.numpy()) == payload
    _rejected(gf, tmp_path, Q4K_HEAD + payload[:287])


# ------------------------------------------------------------------------------------------------ 6. big-endian, duplicates
BIG_ENDIAN = H("""
47 47 55 46                                   # the magic is a byte sequence: the same in both byte orders
00 00 00 03                                   # version 3, big-endian
00 00 00 00 00 00 00 00                       # no tensors
00 00 00 00 00 00 00 01                       # one pair
00 00 00 00 00 00 00 01  6B                   # key "k"
00 00 00 04                                   # UINT32
00 00 00 2A                                   # 42
""")


def test_big_endian_files_are_refused(gf, tmp_path):
    """v3 allows big-endian files (gguf-py byte-swaps them on read).  This reader serves little-endian hosts and files only and says so
    (include/ggq_gguf.h): a byte-swapped version field is GGQ_ERR_FORMAT, the file is never read with the wrong byte order."""
    _rejected(gf, tmp_path, BIG_ENDIAN)


DUPLICATE_KEY = H("""
47 47 55 46  03 00 00 00
00 00 00 00 00 00 00 00
02 00 00 00 00 00 00 00                       # two pairs with the same key
01 00 00 00 00 00 00 00  6B   04 00 00 00   01 00 00 00      # "k" UINT32 1
01 00 00 00 00 00 00 00  6B   04 00 00 00   02 00 00 00      # "k" UINT32 2
""")


def test_duplicate_keys_first_one_wins(gf, tmp_path):
    """gguf-py keeps the first field under its name and files a repeated key under another (name_offset): get_field returns the first."""
    with _open(gf, tmp_path, DUPLICATE_KEY) as f:
        assert f.n_kv == 2 and f.get_field("k").value == 1


def test_absurd_counts_are_refused_before_anything_is_allocated(gf, tmp_path):
    for field_at in (8, 16):                                             # tensor_count, metadata_kv_count
        _rejected(gf, tmp_path, SCALARS[:field_at] + b"\xff" * 8 + SCALARS[field_at + 8:])
    # a string length beyond the file (first key of SCALARS)
    _rejected(gf, tmp_path, SCALARS[:24] + (1 << 40).to_bytes(8, "little") + SCALARS[32:])
    # an array count beyond the file ("a": count field at 253 + 8 + 1 + 4 + 4 = 270)
    _rejected(gf, tmp_path, SCALARS[:270] + (1 << 61).to_bytes(8, "little") + SCALARS[278:])
