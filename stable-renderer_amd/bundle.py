"""Model bundles: a built launch plan written out in relocatable form, so that a host WITHOUT Python or torch can run the model
through the C ABI alone (include/sr_hip.h: sr_model_load / sr_unet_forward / sr_vae_decode; loader: csrc/model.cpp, which also
documents the file layout).

    p = unet.build(B, h, w, inject_idx=[3])            # lowered + tuned once, on the GPU
    export_unet(path, unet, p)                          # -> "SRMODEL1" file: ops, relocations, tensors, io windows
    # any language:  sr_model_load(path, &m);  sr_unet_forward(m, x, t, ctx, out, stream);

Every device pointer inside the plan's ``sr_op`` structs is found through the ctypes field tables of ``_lib`` and replaced by
(tensor id, byte offset) against the storages the plan keeps alive (activations, packed weights, the zero page, the split-K
workspace).  Contents are saved for the weights and for small tensors (norm parameters, index tensors); every other tensor is
an activation buffer and is zero-filled at load (buffers whose padding must read as zero are created zeroed by the plan builder).
"""
import bisect
import ctypes as C
import struct

import torch

from . import _lib as L
from . import ops as O

_MEMBER = {L.OP_IGEMM: ("igemm", L.IgemmArgs), L.OP_GROUPNORM: ("gn", L.GroupNormArgs), L.OP_ATTENTION: ("attn", L.AttentionArgs),
           L.OP_LAYERNORM: ("ln", L._Ln), L.OP_ROW_STATS: ("ln", L._Ln), L.OP_LAYERNORM_GATHER: ("ln", L._Ln), L.OP_NCHW_TO_NHWC: ("cvt", L._Cvt),
           L.OP_NHWC_TO_NCHW: ("cvt", L._Cvt), L.OP_TIMESTEP_EMBED: ("temb", L._Temb), L.OP_SILU: ("ew", L._Ew),
           L.OP_SOFTMAX_ROWS: ("ew", L._Ew), L.OP_GATHER_ROWS: ("gather", L._Gather), L.OP_ADD_SCALED: ("add", L._Add)}
SMALL = 64 << 10          # tensors up to this size keep their contents (index tensors, flags, norm parameters outside the weight dict)


def _pointer_fields(kind):
    """[(byte offset inside sr_op, field name)] of the device pointers of an op of this kind"""
    if kind not in _MEMBER:
        return []
    member, struct_t = _MEMBER[kind]
    base = L.Op.u.offset + getattr(L._OpU, member).offset
    return [(base + getattr(struct_t, name).offset, name) for name, ctype in struct_t._fields_ if ctype is C.c_void_p]


def _storages(tensors):
    """unique storages of the given tensors -> sorted [(start, nbytes, storage)]"""
    seen = {}
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            st = t.untyped_storage()
            seen.setdefault(st.data_ptr(), st)
    return sorted(((ptr, st.nbytes(), st) for ptr, st in seen.items()), key=lambda r: r[0])


def export_bundle(path, plans, io, constants=()):
    """plans: {name: Plan}; io: {name: (tensor, is_output)}; constants: tensors whose contents travel (packed weights)"""
    torch.cuda.synchronize()
    pool = []
    for p in plans.values():
        pool += list(p._keep)
    pool += [t for t, _ in io.values()] + list(constants) + list(O._WS.values()) + list(O._WSC.values()) + list(O._zero_pages.values())
    regions = _storages(pool)
    starts = [r[0] for r in regions]
    const_ptrs = {t.untyped_storage().data_ptr() for t in constants if isinstance(t, torch.Tensor) and t.is_cuda}

    def locate(addr, what):
        i = bisect.bisect_right(starts, addr) - 1
        if i < 0 or addr >= regions[i][0] + max(regions[i][1], 1):
            raise ValueError(f"export_bundle: pointer {addr:#x} ({what}) is not inside any tensor the plan keeps")
        return i, addr - regions[i][0]

    used = {}                                                  # region index -> tensor id, in order of first use

    def tid(i):
        return used.setdefault(i, len(used))
    plan_blobs = []
    for name, p in plans.items():
        relocs, ops = [], bytearray()
        for k in range(p.n):
            op = L.Op()
            C.memmove(C.byref(op), C.byref(p.ops[k]), C.sizeof(L.Op))
            raw = bytearray(bytes(op))
            for off, fname in _pointer_fields(op.kind):
                addr = struct.unpack_from("<Q", raw, off)[0]
                if addr == 0:
                    continue
                i, rel = locate(addr, f"plan {name} op {k} field {fname}")
                relocs.append((k, off, tid(i), rel))
                struct.pack_into("<Q", raw, off, 0)
            ops += raw
        plan_blobs.append((name, p.n, relocs, bytes(ops)))
    io_recs = []
    for name, (t, is_out) in io.items():
        i, rel = locate(t.data_ptr(), f"io {name}")
        io_recs.append((name, tid(i), int(bool(is_out)), rel, t.numel() * t.element_size()))
    order = sorted(used, key=lambda i: used[i])
    # ---- layout: header, tensor table, io, plans, then the data section
    head = 8 + 16 + 16 * len(order) + 56 * len(io_recs)
    for name, n, relocs, ops in plan_blobs:
        head += 16 + 16 + len(ops) + 24 * len(relocs)
    data_off, cursor, payload = [], (head + 255) // 256 * 256, []
    for i in order:
        ptr, nbytes, st = regions[i]
        keep = nbytes > 0 and (ptr in const_ptrs or nbytes <= SMALL)
        if keep:
            view = torch.empty(0, dtype=torch.uint8, device=st.device).set_(st, 0, (nbytes,))
            blob = view.cpu().numpy().tobytes()
            keep = any(blob) or ptr in const_ptrs
        if keep:
            data_off.append(cursor)
            payload.append(blob)
            cursor += (nbytes + 255) // 256 * 256
        else:
            data_off.append(0)
            payload.append(None)
    with open(path, "wb") as f:
        f.write(b"SRMODEL1" + struct.pack("<IIII", 1, len(order), len(io_recs), len(plan_blobs)))
        for i, off in zip(order, data_off):
            f.write(struct.pack("<QQ", regions[i][1], off))
        for name, t_id, is_out, rel, nb in io_recs:
            f.write(name.encode()[:31].ljust(32, b"\0") + struct.pack("<IIQQ", t_id, is_out, rel, nb))
        for name, n, relocs, ops in plan_blobs:
            f.write(name.encode()[:15].ljust(16, b"\0") + struct.pack("<IIII", n, len(relocs), C.sizeof(L.Op), 0) + ops)
            for k, off, t_id, rel in relocs:
                f.write(struct.pack("<IIIIQ", k, off, t_id, 0, rel))
        assert f.tell() == head, (f.tell(), head)
        for off, blob, i in zip(data_off, payload, order):
            if blob is not None:
                f.seek(off)
                f.write(blob)
        f.truncate(max(cursor, head))
    return dict(tensors=len(order), bytes=cursor, saved=sum(b is not None for b in payload))


def export_unet(path, unet, built):
    """``built`` = UNet.build(...) (view-sharded schedules are not bundled: their collectives belong to the host)"""
    if built.get("schedule"):
        raise NotImplementedError("a view-sharded plan is a schedule of segments and collectives: bundle the unsharded plan")
    io = {"x": (built["x"], False), "t": (built["t"], False), "ctx": (built["ctx"], False), "out": (built["out"], True)}
    if built.get("y") is not None:
        io["y"] = (built["y"], False)
    if built.get("inject") is not None:
        io["inject"] = (built["inject"], False)
    if built.get("inject_err") is not None:                 # the device flag an out-of-range injected index raises (sr_gather_rows)
        io["inject_err"] = (built["inject_err"], True)
    return export_bundle(path, {"prologue": built["prologue"], "step": built["step"]}, io, constants=list(unet.w.values()))


def export_vae(path, vae, built):
    """``built`` = VAEDecoder.build(n, h, w)"""
    return export_bundle(path, {"step": built["plan"]}, {"z": (built["z"], False), "img": (built["img"], True)}, constants=list(vae.w.values()))
