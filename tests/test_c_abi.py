"""The C-ABI library loads without a GPU and exports every symbol include/mmt_attn.h declares;
argument errors are reported through return codes + mmt_last_error (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
  import __graft_entry__  # noqa: F401  (sets sys.path)
  from mmt_amd import _lib
  _lib.build()
  return _lib


def test_header_symbols_are_exported(lib):
  header = open(os.path.join(ROOT, 'include', 'mmt_attn.h')).read()
  layer_header = open(os.path.join(ROOT, 'include', 'mmt_layer.h')).read()
  declared = set(re.findall(r'\b(mmt_[a-z_]+)\s*\(', header + layer_header))
  assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
  L = lib.lib()
  for name in declared:
    assert hasattr(L, name), name
  assert L.mmt_abi_version() == 4
  assert int(re.search(r'#define MMT_ABI_VERSION (\d+)', header).group(1)) == 4


def test_struct_layout_matches_header(lib):
  # field order / widths of the ctypes mirrors against the header text
  header = open(os.path.join(ROOT, 'include', 'mmt_attn.h')).read()
  body = header[header.index('typedef struct mmt_mask_desc'):header.index('} mmt_mask_desc;')]
  names = re.findall(r'\b(?:int32_t\*?|const int32_t\*)\s+(\w+);', body)
  assert names == [f[0] for f in lib.MaskDesc._fields_]
  body = header[header.index('typedef struct mmt_attn_desc'):header.index('} mmt_attn_desc;')]
  names = re.findall(r'\b(\w+)(?:\[3\])?;', body)
  flat = []
  for line in body.splitlines():
    m = re.match(r'\s*(?:int32_t|int64_t|float|uint32_t\*?|uint64_t|const uint64_t\*|mmt_mask_desc)\s+([^;]+);', line)
    if m:
      flat += [x.strip().split('[')[0] for x in m.group(1).split(',')]
  assert flat == [f[0] for f in lib.AttnDesc._fields_]
  assert ctypes.sizeof(lib.MaskDesc) == 8 + 7 * 4 + 4 + 8     # pointer + 7 ints (padded to 8) + the ABI-2 index pointer


def test_layer_descriptor_layouts_match_the_header(lib):
  """mmt_layer.h's descriptors (ABI 4 appended the device-resident step scalars to each) against their ctypes mirrors."""
  header = open(os.path.join(ROOT, 'include', 'mmt_layer.h')).read()
  for cname, mirror in (('mmt_rows_desc', lib.RowsDesc), ('mmt_embed_desc', lib.EmbedDesc), ('mmt_adamw_desc', lib.AdamwDesc)):
    body = header[header.index(f'typedef struct {cname}'):header.index(f'}} {cname};')]
    flat = []
    for line in body.splitlines():
      m = re.match(r'\s*(?:int32_t|int64_t|float|uint32_t|uint64_t|const uint64_t\*|const float\*)\s+([^;]+);', line)
      if m:
        flat += [x.strip() for x in m.group(1).split(',')]
    assert flat == [f[0] for f in mirror._fields_], cname


def test_the_library_reads_no_environment_variable_and_keeps_no_step_state():
  """ABI 4: kernel-selection switches and the device-resident step scalars travel in the descriptors.  No getenv in the
  product build of the library (the -DMMT_STAMP diagnostic build keeps its MMT_DBG_* reads), no registration call."""
  csrc = os.path.join(ROOT, 'multimodal-long-transformer-2021_amd', 'csrc')
  for name in os.listdir(csrc):
    if not name.endswith(('.hip', '.h')):
      continue
    src = open(os.path.join(csrc, name)).read()
    src = re.sub(r'#ifdef MMT_STAMP.*?#endif', '', src, flags=re.S)
    assert 'getenv' not in src, name
    assert 'g_dropout_epoch' not in src and 'g_adamw_hyper' not in src, name
  header = open(os.path.join(ROOT, 'include', 'mmt_attn.h')).read()
  assert not re.search(r'\bint\s+mmt_set_step_scalars\s*\(', header)


def test_integration_document_stub_matches_the_binding(lib):
  """The ctypes stub printed in INTEGRATION.md section 2 is what a maintainer copies: its field lists and ctypes
  types must be the ones of mmt_amd/_lib.py (which test_struct_layout_matches_header ties to the header), and the
  structs built from the document must have the binding's sizes (MaskDesc 48 bytes since ABI 2)."""
  doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  sec = doc[doc.index('## 2. ctypes stub'):doc.index('## 3.')]
  code = sec[sec.index('```python') + len('```python'):]
  code = code[:code.index('```')]
  # execute the two class definitions only (no library load, no call)
  defs = code[code.index('class MaskDesc'):code.index('lib.mmt_attn_fwd.restype')]
  ns = {'ctypes': ctypes}
  exec(defs, ns)
  for name in ('MaskDesc', 'AttnDesc'):
    doc_fields = [(f[0], f[1]) for f in ns[name]._fields_]
    lib_fields = [(f[0], f[1]) for f in getattr(lib, name)._fields_]
    assert [f[0] for f in doc_fields] == [f[0] for f in lib_fields], name
    for (n, t_doc), (_, t_lib) in zip(doc_fields, lib_fields):
      if n == 'mask':
        assert ctypes.sizeof(t_doc) == ctypes.sizeof(t_lib)
      else:
        assert t_doc is t_lib or (ctypes.sizeof(t_doc), t_doc._type_ if hasattr(t_doc, '_type_') else None) == \
            (ctypes.sizeof(t_lib), t_lib._type_ if hasattr(t_lib, '_type_') else None), (name, n)
    assert ctypes.sizeof(ns[name]) == ctypes.sizeof(getattr(lib, name)), name
  assert ctypes.sizeof(ns['MaskDesc']) == 48


def test_sanitizer_build_of_the_shim():
  """`make asan` (csrc/Makefile): the host side of the C-ABI shim under AddressSanitizer + UBSan, the launchers
  replaced by stand-ins that check every derived workspace pointer, driven by a C program through include/mmt_attn.h
  (SURVEY.md section 5).  CPU only."""
  import shutil
  import subprocess
  if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
    pytest.skip('no hipcc on this machine')
  csrc = os.path.join(ROOT, 'multimodal-long-transformer-2021_amd', 'csrc')
  res = subprocess.run(['make', '-s', '-C', csrc, 'asan'], capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  assert 'asan driver ok' in res.stdout and 'ERROR: AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr


def test_argument_errors_without_gpu(lib):
  L = lib.lib()
  d = lib.AttnDesc()
  d.B, d.S, d.N, d.D = 1, 64, 1, 32
  assert L.mmt_workspace_bytes(d) == 0
  assert b'head size 64' in L.mmt_last_error()
  d.D = 64
  d.R = 129                                                # tables are built up to 128 ids wide
  assert L.mmt_workspace_bytes(d) == 0 and b'[0,128]' in L.mmt_last_error()
  d.R = 0
  d.dtype = lib.MMT_BF16
  for arr in (d.q_stride, d.k_stride, d.v_stride, d.o_stride):
    arr[:] = (64 * 64, 64, 64)
  d.scale, d.mask_value = 0.125, -10000.0
  d.mask.local_radius = 8
  d.mask.n_global, d.mask.global_start = 4, 62            # range leaves the sequence
  assert L.mmt_attn_fwd(d, 1, 1, 1, None, None, None, None, 1, None, None, 0, None) == -1
  assert b'global range' in L.mmt_last_error()
  d.mask.global_start = 10
  assert L.mmt_workspace_bytes(d) > 0
  assert L.mmt_attn_fwd(d, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
  d.q_stride[1] = 60                                       # breaks 16-byte alignment
  assert L.mmt_attn_fwd(d, 1, 1, 1, None, None, None, None, 1, None, None, 0, None) == -1
  d.q_stride[1] = 64
  # a structured call that needs workspace but gets none
  assert L.mmt_attn_fwd(d, 1, 1, 1, None, None, None, None, 1, None, None, 0, None) == -3
  assert b'workspace' in L.mmt_last_error()
  m = lib.MaskDesc()
  m.id_mode, m.max_dist, m.patches_per_row, m.core_layers = 2, 3, 0, 1
  assert L.mmt_side_inputs(m, 1, 16, None, None, 0, None, None, None, None) == -1
  assert b'`num_patch_per_row` must be positive.' in L.mmt_last_error()


def test_no_cpu_fallback(lib):
  import torch
  import mmt_amd
  x = torch.zeros(1, 32, 1, 64)
  with pytest.raises(RuntimeError, match='GPU only'):
    mmt_amd.relative_attention_forward(x, x, x)
