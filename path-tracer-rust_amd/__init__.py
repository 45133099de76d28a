"""Python binding of libptrace_hip.so (C ABI in include/ptrace.h) for bench.py, smoke() and the rank harness.

Import with importlib.import_module("path-tracer-rust_amd").  This module is plumbing only: it loads the
in-tree shared library (and fails loudly if it is missing — there is no Python or CPU fallback), mirrors the
POD structs, and provides the band partition / framebuffer gather used when one process per GPU renders a
contiguous band of the reference's `pixels` vector (src/render/mod.rs:1017-1024: pixels are independent).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB") or os.path.join(_HERE, "libptrace_hip.so")  # PT_LIB: A/B builds when tuning

f3 = C.c_float * 3


class PtraceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ptrace error %d: %s" % (code, msg))
        self.code = code


class pt_camera(C.Structure):
    _fields_ = [("position", f3), ("direction", f3), ("focal_length", C.c_float),
                ("sensor_width", C.c_float), ("aspect_ratio", C.c_float)]


class pt_triangle(C.Structure):
    _fields_ = [("a", f3), ("b", f3), ("c", f3)]


class pt_object(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("position", f3), ("radius", C.c_float), ("color", f3),
                ("emission", f3), ("reflect_type", C.c_uint32), ("tri_offset", C.c_uint32),
                ("tri_count", C.c_uint32), ("bs_center", f3), ("bs_radius", C.c_float)]


class pt_config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("backend", C.c_uint32),
                ("seed", C.c_uint64), ("idx_begin", C.c_uint32), ("idx_end", C.c_uint32),
                ("rays_per_pass", C.c_uint32), ("flags", C.c_uint32), ("chunk_pixels", C.c_uint32),
                ("chunk_first", C.c_uint32), ("chunk_step", C.c_uint32), ("progress_ms", C.c_uint32)]


class pt_stats(C.Structure):
    _fields_ = [("ray_bounces", C.c_uint64), ("samples", C.c_uint64), ("intersect_rays", C.c_uint64),
                ("intersect_launches", C.c_uint32), ("passes", C.c_uint32), ("ms_total", C.c_double),
                ("ms_device", C.c_double), ("ms_intersect", C.c_double)]


PT_OK, PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_CANCELLED = 0, -1, -2, -3, -4
BACKEND_WAVEFRONT, BACKEND_MEGAKERNEL = 0, 1
PT_FLAG_NO_BVH, PT_FLAG_SEPARATE_KERNELS = 1, 2
BACKENDS = {"wavefront": BACKEND_WAVEFRONT, "megakernel": BACKEND_MEGAKERNEL}

_lib = None


def lib():
    """The loaded C ABI.  Raises if the HIP library has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C %s` (hipcc, gfx950); there is no fallback path"
                          % (LIB_PATH, _HERE))
    L = C.CDLL(LIB_PATH)
    L.pt_version.restype = C.c_char_p
    L.pt_build_flags.restype = C.c_char_p
    L.pt_kernel_isa_hash.restype = C.c_char_p
    L.pt_last_error.restype = C.c_char_p
    L.pt_device_count.restype = C.c_int
    L.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pt_ctx_destroy.argtypes = [C.c_void_p]
    L.pt_ctx_destroy.restype = None
    L.pt_ctx_set_scene.argtypes = [C.c_void_p, C.POINTER(pt_camera), C.POINTER(pt_object), C.c_uint32,
                                   C.POINTER(pt_triangle), C.c_uint32]
    L.pt_ctx_render.argtypes = [C.c_void_p, C.POINTER(pt_config), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(pt_stats)]
    L.pt_ctx_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.pt_ctx_pass_kernel.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_ctx_pass_kernel.restype = C.c_char_p
    L.pt_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.pt_scene_free.argtypes = [C.c_void_p]
    L.pt_scene_free.restype = None
    L.pt_scene_id.argtypes = [C.c_void_p]
    L.pt_scene_id.restype = C.c_char_p
    L.pt_scene_camera.argtypes = [C.c_void_p]
    L.pt_scene_camera.restype = C.POINTER(pt_camera)
    L.pt_scene_objects.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.pt_scene_objects.restype = C.POINTER(pt_object)
    L.pt_scene_triangles.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.pt_scene_triangles.restype = C.POINTER(pt_triangle)
    L.pt_write_ppm.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p,
                               C.c_uint64]
    L.pt_image_hash.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    L.pt_image_hash.restype = C.c_uint64
    L.pt_comm_unique_id.argtypes = [C.c_char_p]
    L.pt_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    L.pt_comm_destroy.argtypes = [C.c_void_p]
    L.pt_comm_destroy.restype = None
    L.pt_comm_gather_frame.argtypes = [C.c_void_p, C.POINTER(pt_config), C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != PT_OK:
        raise PtraceError(rc, lib().pt_last_error().decode())


class Scene:
    """SceneDescriptor::load + to_data through the C ABI (pt_scene_load)."""

    def __init__(self, path, base_dir=None):
        L = lib()
        base_dir = base_dir or os.path.dirname(os.path.dirname(os.path.abspath(path)))
        self._h = C.c_void_p()
        _check(L.pt_scene_load(path.encode(), base_dir.encode(), C.byref(self._h)))
        n, m = C.c_uint32(), C.c_uint32()
        self.objects = L.pt_scene_objects(self._h, C.byref(n))
        self.triangles = L.pt_scene_triangles(self._h, C.byref(m))
        self.n_objects, self.n_triangles = n.value, m.value
        self.camera = L.pt_scene_camera(self._h)
        self.id = L.pt_scene_id(self._h).decode()

    def close(self):
        if self._h:
            lib().pt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """pt_ctx: one GPU, one scene, the ray streams."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().pt_ctx_create(device, C.byref(self._h)))

    def set_scene(self, scene):
        _check(lib().pt_ctx_set_scene(self._h, scene.camera, scene.objects, scene.n_objects, scene.triangles,
                                      scene.n_triangles))

    def pass_kernel(self, separate_kernels=False):
        """Name of the kernel a wavefront pass of this scene launches (see pt_ctx_pass_kernel)."""
        n = lib().pt_ctx_pass_kernel(self._h, PT_FLAG_SEPARATE_KERNELS if separate_kernels else 0)
        return n.decode() if n else None

    def set_profiling(self, on):
        _check(lib().pt_ctx_set_profiling(self._h, 1 if on else 0))

    def render(self, out_ptr, width, height, spp, seed=1, backend="wavefront", band=None, rays_per_pass=0,
               stream=None, chunks=None, pipelines=1, separate_kernels=False):
        """Render band [begin,end) (default whole frame) — or, with chunks=(chunk_pixels, first, step), this rank's
        interleaved chunks of it — into device memory at out_ptr (owned pixels * 3 floats)."""
        cfg = pt_config(width, height, spp, BACKENDS[backend], seed, 0, 0, rays_per_pass,
                        (((pipelines & 15) << 8) if pipelines > 1 else 0) | (PT_FLAG_SEPARATE_KERNELS if separate_kernels else 0))
        if band is not None:
            cfg.idx_begin, cfg.idx_end = band
        if chunks is not None:
            cfg.chunk_pixels, cfg.chunk_first, cfg.chunk_step = chunks
        st = pt_stats()
        _check(lib().pt_ctx_render(self._h, C.byref(cfg), C.c_void_p(out_ptr), C.c_void_p(stream or 0), None, None,
                                   None, C.byref(st)))
        return st

    def close(self):
        if self._h:
            lib().pt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """pt_comm: the RCCL framebuffer gather behind the C ABI (include/ptrace.h).  `unique_id()` on rank 0, the 128
    bytes carried to the other ranks by the host's own means (bench.py: the torch.distributed store), then every rank
    constructs Comm(device, rank, world, id) collectively."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(lib().pt_comm_unique_id(buf))
        return buf.raw

    def __init__(self, device, rank, world, ident):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        _check(lib().pt_comm_create(device, rank, world, C.create_string_buffer(ident, Comm.ID_BYTES), C.byref(self._h)))

    def gather_frame(self, local_ptr, frame_ptr, width, height, chunk_pixels, band=None, stream=None):
        """Every rank gets the whole band in device memory at frame_ptr (one ncclAllGather + the un-permute kernel)."""
        cfg = pt_config(width, height, 1, 0, 0, 0, 0, 0, 0)
        if band is not None:
            cfg.idx_begin, cfg.idx_end = band
        cfg.chunk_pixels = chunk_pixels
        _check(lib().pt_comm_gather_frame(self._h, C.byref(cfg), C.c_void_p(local_ptr), C.c_void_p(frame_ptr),
                                          C.c_void_p(stream or 0)))

    def close(self):
        if self._h:
            lib().pt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_flags():
    """The back-end (-mllvm) switches the library was built with (the Makefile drops the ones the compiler rejects)."""
    return lib().pt_build_flags().decode()


def kernel_isa_hash():
    """Hash of the device assembly the loaded library's kernels were built from (profiles name the one they measured)."""
    return lib().pt_kernel_isa_hash().decode()


# the -mllvm set the Makefile asks for (instruction placement only; worth 3.5 % on mesh.json): a library built without some of
# them - its compiler did not know them - still renders the same images
# (two sets, "general | flat: k_pass_cand without walks" - that kernel is a translation unit of its own, pt_kernels_flat.hip)
BUILD_FLAGS_WANTED = ("-amdgpu-sched-strategy=max-ilp", "-disable-machine-licm", "-disable-machine-sink")
BUILD_FLAGS_WANTED_FLAT = ("-amdgpu-sched-strategy=iterative-minreg", "-disable-machine-licm")


def build_flags_complete():
    general, _, flat = build_flags().partition("| flat:")
    return all(f in general for f in BUILD_FLAGS_WANTED) and all(f in flat for f in BUILD_FLAGS_WANTED_FLAT)


def image_hash(frame):
    """Image.hash (mod.rs:916-926) of a [pixels, 3] float32 torch tensor (any device): pt_image_hash over its bits."""
    host = frame.detach().to("cpu").contiguous()
    return int(lib().pt_image_hash(C.cast(host.data_ptr(), C.POINTER(C.c_float)), host.numel()))


def band_for_rank(npix, rank, world):
    """Contiguous band of framebuffer indices for `rank` of `world`: [rank*npix/world, (rank+1)*npix/world).
    idx = (H-1-y)*W + x is row-major (mod.rs:805-806), so a band is one contiguous slice of the image."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return (npix * rank) // world, (npix * (rank + 1)) // world


def chunk_counts(npix, world, chunk_pixels):
    """Pixels owned by each rank under the interleaved partition (chunk c belongs to rank c % world), by arithmetic."""
    n_chunks = (npix + chunk_pixels - 1) // chunk_pixels
    last = npix - (n_chunks - 1) * chunk_pixels  # pixels of the last (possibly short) chunk
    counts = []
    for r in range(world):
        mine = (n_chunks - r + world - 1) // world if n_chunks > r else 0
        c = mine * chunk_pixels
        if mine and (n_chunks - 1) % world == r:
            c -= chunk_pixels - last
        counts.append(c)
    return counts


def chunk_owner_map(npix, world, chunk_pixels):
    """Interleaved partition spelled out: chunk c (chunk_pixels consecutive framebuffer indices) belongs to rank
    c % world.  Returns (counts, index) with index[r] = the framebuffer indices of rank r's pixels in the order
    pt_ctx_render writes them.  O(npix * world): the explicit form the tests hold the arithmetic against."""
    import torch
    idx = torch.arange(npix, dtype=torch.int64)
    owner = (idx // chunk_pixels) % world
    index = [idx[owner == r] for r in range(world)]
    return [int(i.numel()) for i in index], index


def gather_chunks(local, npix, rank, world, chunk_pixels, dist=None, force_collective=False):
    """One all-gather of the per-rank chunk buffers ([owned pixels, 3]) and the permutation back to framebuffer
    order.  Equal shares (npix a multiple of world*chunk_pixels) use all_gather_into_tensor + a strided view copy;
    ragged shares pad every rank to whole chunks and copy rank r's chunks to the frame's chunks r, r+world, ...
    (one strided copy per rank: O(npix) bytes, no index tensors).  world == 1 is the identity unless
    force_collective asks for the collective anyway (hardware check of the RCCL path on a one-GPU box)."""
    import torch
    if world == 1 and not force_collective:
        return local
    if dist is None:
        import torch.distributed as dist
    counts = chunk_counts(npix, world, chunk_pixels)
    assert local.shape[0] == counts[rank]
    if len(set(counts)) == 1 and npix % (world * chunk_pixels) == 0:
        flat = torch.empty((npix, 3), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, local.contiguous())
        rounds = npix // (world * chunk_pixels)
        # flat is [rank][round][chunk_pixels]; the frame is [round][rank][chunk_pixels]
        return flat.view(world, rounds, chunk_pixels, 3).permute(1, 0, 2, 3).reshape(npix, 3)
    n_chunks = (npix + chunk_pixels - 1) // chunk_pixels
    per_rank = [len(range(r, n_chunks, world)) for r in range(world)]  # chunks of each rank (the last may be short)
    m = max(per_rank) * chunk_pixels
    padded = torch.zeros((m, 3), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    frame = torch.empty((n_chunks, chunk_pixels, 3), dtype=local.dtype, device=local.device)
    for r in range(world):
        if per_rank[r]:
            frame[r::world] = parts[r][: per_rank[r] * chunk_pixels].view(per_rank[r], chunk_pixels, 3)
    return frame.view(n_chunks * chunk_pixels, 3)[:npix]


def gather_bands(local, npix, rank, world, dist=None):
    """All-gather the per-rank bands (torch tensors of shape [band_pixels, 3]) into the full [npix, 3] image.
    One collective: all_gather_into_tensor when the bands are equal, all_gather on padded bands otherwise."""
    import torch
    if world == 1:
        return local
    if dist is None:
        import torch.distributed as dist
    sizes = [band_for_rank(npix, r, world)[1] - band_for_rank(npix, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        full = torch.empty((npix, 3), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous())
        return full
    m = max(sizes)
    padded = torch.zeros((m, 3), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)
