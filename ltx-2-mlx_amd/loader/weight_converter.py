"""safetensors -> HBM loader for the DiT (reference LTX_2_MLX/loader/weight_converter.py:277-446,527-553,
loader/fp8_loader.py:14-130).

Same entry points and keyword meaning as the reference.  What differs is where the bytes go: each
tensor is read once from the (memory-mapped) safetensors file and copied straight to HBM; fp8
(`float8_e4m3fn` + per-tensor `<key>.weight_scale`) weights are uploaded as raw bytes and dequantised
ON the GPU (`ltx2_dequant_fp8_e4m3fn`: f32(fp8) * scale -> bf16, the reference's arithmetic at
weight_converter.py:391-395 with bf16 instead of fp16 as the resident dtype).  The reference's
torch -> numpy -> MLX double copy (weight_converter.py:383-431) does not exist here.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import kernels as K

PREFIX = "model.diffusion_model."


def convert_pytorch_key(key: str, include_audio: bool = False) -> Optional[str]:
    """Key filter of convert_pytorch_key_to_mlx (weight_converter.py:277-315).  The engine keeps the
    checkpoint's own names (to_out.0, ff.net.0.proj, ff.net.2), so only the skip rules apply."""
    if not include_audio and ("av_ca" in key or "a2v" in key or "audio" in key.lower()):
        return None
    if "video_embeddings_connector" in key or "audio_embeddings_connector" in key:
        return None                      # text-encoder weights, loaded elsewhere in the reference
    return key


from ..model.transformer import FP8_RESIDENT_KEYS, Fp8Weight  # noqa: E402


_ST_DTYPES = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64, "F8_E4M3": torch.float8_e4m3fn,
              "F8_E5M2": torch.float8_e5m2, "U8": torch.uint8, "I8": torch.int8, "I16": torch.int16, "I32": torch.int32, "I64": torch.int64,
              "BOOL": torch.bool}


class SafetensorsStream:
    """safetensors file -> HBM through pinned staging: the file is memory-mapped once, tensors are packed back to back into one of two
    pinned host buffers (`stage_bytes` each) and every full buffer leaves as ONE asynchronous copy on a dedicated copy stream into a
    device arena, while the host fills the other buffer (mmap page-cache reads overlap the PCIe/xGMI transfer).  The tensors handed out
    are views of the arenas.  The reference's loader copies torch -> numpy -> MLX per tensor (loader/weight_converter.py:383-431); the
    round-2 loader here did one pageable `t.to(device)` per tensor.

        with SafetensorsStream(path, device) as st:
            for name in st.keys(): ...
            tensors = st.load(names)          # dict name -> device tensor (checkpoint dtype)

    `stats`: bytes moved and seconds spent, for the load-time figure bench.py reports."""

    def __init__(self, path: str, device, stage_bytes: int = 256 << 20):
        import json
        import struct
        self.path, self.device = path, torch.device(device)
        with open(path, "rb") as f:
            (hlen,) = struct.unpack("<Q", f.read(8))
            self.header = json.loads(f.read(hlen))
        self.metadata = self.header.pop("__metadata__", {}) or {}
        self.data0 = 8 + hlen
        self._mm = None
        self.stage_bytes = stage_bytes
        self.stats = {"bytes": 0, "seconds": 0.0}

    def __enter__(self):
        import os
        self._mm = torch.from_file(self.path, shared=False, size=os.path.getsize(self.path), dtype=torch.uint8)
        return self

    def __exit__(self, *exc):
        self._mm = None
        return False

    def keys(self):
        return list(self.header.keys())

    def info(self, name):
        e = self.header[name]
        return _ST_DTYPES[e["dtype"]], tuple(e["shape"]), e["data_offsets"]

    def load(self, names):
        """names -> {name: device tensor}.  One pass in file order (sequential page-cache reads)."""
        import time
        t0 = time.perf_counter()
        order = sorted(names, key=lambda n: self.header[n]["data_offsets"][0])
        out = {}
        if self.device.type != "cuda":
            for n in order:
                dt, shape, (a, b) = self.info(n)
                out[n] = self._mm[self.data0 + a: self.data0 + b].clone().view(dt).reshape(shape)
            return out
        from concurrent.futures import ThreadPoolExecutor
        copy_stream = torch.cuda.Stream(device=self.device)
        # The arenas are allocated on the CURRENT stream and filled on copy_stream: the caching allocator may hand back a block that
        # kernels still queued on the current stream read (the previous model's weights, a load_state_dict temporary), so the copies
        # must start after everything enqueued there so far; record_stream below keeps a freed arena away from reuse
        # until the copy stream is done with it.
        copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        stage = [torch.empty(self.stage_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        pool = ThreadPoolExecutor(max_workers=8)
        PIECE = 16 << 20

        def host_copy(dst, src):
            """mmap -> pinned memcpy in 16-MiB pieces over 8 threads (one thread moves ~6 GB/s; torch's copy_ drops the GIL)."""
            n = dst.numel()
            if n <= PIECE:
                dst.copy_(src)
                return []
            return [pool.submit(dst[o: o + PIECE].copy_, src[o: o + PIECE]) for o in range(0, n, PIECE)]

        free_ev = [None, None]
        # batches of tensors that fit one staging buffer (a tensor larger than the buffer goes alone, in slices)
        batches, cur, used = [], [], 0
        for n in order:
            a, b = self.header[n]["data_offsets"]
            nb = (b - a + 255) // 256 * 256
            if cur and used + nb > self.stage_bytes:
                batches.append(cur)
                cur, used = [], 0
            cur.append((n, used))
            used += nb
        if cur:
            batches.append(cur)
        total = 0
        for bi, batch in enumerate(batches):
            last_n, last_off = batch[-1]
            la, lb = self.header[last_n]["data_offsets"]
            span = last_off + (lb - la)
            arena = torch.empty(max(span, 1), dtype=torch.uint8, device=self.device)
            arena.record_stream(copy_stream)
            done = 0
            while done < span:                           # (one pass unless a single tensor exceeds the staging buffer)
                k = bi % 2 if span <= self.stage_bytes else (done // self.stage_bytes) % 2
                if free_ev[k] is not None:
                    free_ev[k].synchronize()             # the previous copy out of this buffer has finished
                n_here = min(self.stage_bytes, span - done)
                futs = []
                if span <= self.stage_bytes:
                    for n, off in batch:
                        a, b = self.header[n]["data_offsets"]
                        futs += host_copy(stage[k][off: off + b - a], self._mm[self.data0 + a: self.data0 + b])
                else:                                    # oversize single tensor: stream it through in slices
                    a, _ = self.header[last_n]["data_offsets"]
                    futs += host_copy(stage[k][:n_here], self._mm[self.data0 + a + done: self.data0 + a + done + n_here])
                for fu in futs:
                    fu.result()
                with torch.cuda.stream(copy_stream):
                    arena[done: done + n_here].copy_(stage[k][:n_here], non_blocking=True)
                    free_ev[k] = torch.cuda.Event()
                    free_ev[k].record(copy_stream)
                done += n_here
            for n, off in batch:
                dt, shape, (a, b) = self.info(n)
                out[n] = arena[off: off + b - a].view(dt).reshape(shape)
            total += span
        copy_stream.synchronize()
        pool.shutdown()
        torch.cuda.current_stream(self.device).wait_stream(copy_stream)
        self.stats["bytes"] += total
        self.stats["seconds"] += time.perf_counter() - t0
        return out


def is_fp8_checkpoint(weights_path: str) -> bool:
    """True if any `<key>.weight_scale` entry exists (fp8_loader.py:133-150)."""
    return any(k.endswith(".weight_scale") for k in SafetensorsStream(weights_path, "cpu").keys())


def load_transformer_weights(model, weights_path: str, strict: bool = False, use_fp8: bool = False,
                             include_audio: bool = False, streaming: bool = True, target_dtype: str = "float16",
                             lora_configs=None, fp8_resident: bool = False):
    """Load `model.diffusion_model.*` tensors into an LTXModel (weight_converter.py:318-446).

    use_fp8: dequantise fp8 weights with their `weight_scale` (ignored `input_scale`, fp8_loader.py:87-97);
    include_audio: keep audio / av_ca / a2v keys (AudioVideo model); `streaming` and `target_dtype` (reference default
    "float16") are accepted for signature compatibility (loading always streams; the resident dtype is bf16);
    fp8_resident (MI355X addition, BASELINE config 3): the video stream's attention / feed-forward projections stay
    float8_e4m3fn + scale in HBM (half the bytes) and are expanded inside the GEMM -- bit-identical to dequantising at load;
    lora_configs: LoRAConfig list fused into the checkpoint-keyed weights on the GPU before they are packed
    (reference loader/lora_loader.py:129-194)."""
    dev = model.device
    sd: Dict[str, torch.Tensor] = {}
    n_fp8 = n_res = 0
    if fp8_resident and lora_configs:
        raise NotImplementedError("LoRA fusion needs dequantised weights: drop fp8_resident")
    with SafetensorsStream(weights_path, dev) as f:
        keys = f.keys()
        scales = {}
        if use_fp8:
            sk = [k for k in keys if k.endswith(".weight_scale")]
            for k, t in f.load(sk).items():
                scales[k[:-len("_scale")]] = float(t.float().item())
        want = {}
        for full in keys:
            if not full.startswith(PREFIX) or full.endswith("_scale"):
                continue
            key = convert_pytorch_key(full[len(PREFIX):], include_audio=include_audio)
            if key is not None:
                want[full] = key
        for full, t in f.load(list(want)).items():
            key = want[full]
            if full in scales:
                if t.dtype != torch.float8_e4m3fn:
                    raise ValueError(f"{full}: has a weight_scale but dtype {t.dtype}, expected float8_e4m3fn")
                if fp8_resident and FP8_RESIDENT_KEYS.match(key) and t.shape[0] % 256 == 0 and t.shape[1] % 128 == 0 and t.shape[1] >= 256:
                    sd[key] = Fp8Weight(t.view(torch.uint8), scales[full])
                    n_res += 1
                else:
                    sd[key] = K.dequant_fp8(t.view(torch.uint8), scales[full], dtype=model.compute_dtype)
                n_fp8 += 1
            elif t.dtype == torch.float8_e4m3fn:          # fp8 without a scale (weight_converter.py:399-401)
                sd[key] = K.dequant_fp8(t.view(torch.uint8), 1.0, dtype=model.compute_dtype)
                n_fp8 += 1
            else:
                sd[key] = t
        stats = dict(f.stats)
    if lora_configs:
        from .lora_loader import fuse_lora_into_weights
        sd = fuse_lora_into_weights(sd, lora_configs)
    model.load_state_dict(sd, strict=strict)
    gbs = stats["bytes"] / max(stats["seconds"], 1e-9) / 1e9
    print(f"  loaded {len(sd)} transformer tensors ({n_fp8} fp8, {n_res} of them kept fp8-resident) from {weights_path}: "
          f"{stats['bytes'] / 1e9:.2f} GB file -> HBM in {stats['seconds']:.2f} s ({gbs:.1f} GB/s, pinned staging)")
    return stats


def load_av_transformer_weights(model, weights_path: str, strict: bool = False, use_fp8: bool = False,
                                target_dtype: str = "float16", lora_configs=None, fp8_resident: bool = False):
    """load_transformer_weights(include_audio=True) (weight_converter.py:527-553).  `lora_configs` / `fp8_resident`: as there (the
    reference fuses LoRA into whichever transformer is loaded, scripts/generate.py:1186-1202)."""
    return load_transformer_weights(model, weights_path, strict=strict, use_fp8=use_fp8, include_audio=True, target_dtype=target_dtype,
                                    lora_configs=lora_configs, fp8_resident=fp8_resident)
