// vmem.hip — growable device arenas on HIP virtual memory (see include/rarc.h: rarc_vmem_*).
//
// index.add in the reference appends (faiss grows its std::vector: VectorStore_Faiss.py:199-202).  A row buffer in HBM that
// grows by allocate-bigger-and-copy holds old + new at once — up to 3x the live rows — so a store past a third of the
// part's memory could not grow.  An arena owns a range of virtual addresses as large as it may ever get (costs no memory)
// and backs it slab by slab (hipMemCreate + hipMemMap + hipMemSetAccess) as rows arrive: the base pointer never moves,
// nothing is copied, the kernels see one contiguous buffer as before, and the peak footprint is the live rows rounded up
// to a slab.
//
// How the address space is managed is dictated by what this runtime (ROCm 7.0.2 / HIP 7.0.51831) does, measured with
// tools/vmem_probe.py and by running the GPU suite in sequence:
//   * inside one reservation the second mapped piece fixes the size of all later ones (2,30 MiB ok; 2,8,8,8 ok, then 6 ->
//     hipErrorInvalidValue; 2,2 then 4 -> invalid);
//   * an address range that was hipMemAddressFree'd and handed out again by a later hipMemAddressReserve — another size,
//     another phase — maps without error and then loses writes / reads back stale data (tests passed alone and failed
//     in sequence; a reservation per arena, however aligned, did not cure it);
//   * hipMemAddressReserve ignores its alignment argument (2 MiB-aligned ranges whatever is asked).
//   * even inside ONE never-freed reservation with uniform pieces, an address that was unmapped and then mapped again
//     (a destroyed index's range handed to the next one) lost data now and then: rows written by kernels read back as
//     zeros ~120 tests into the suite, each test passing alone.
// The sequences the runtime answers differently from box to box (tools/vmem_probe.py: "2 then 8" maps on one box and
// fails on the next), so the design uses nothing but what never misbehaved: ONE reservation that is never freed, pieces of
// ONE size on one grid, and NO ADDRESS EVER MAPPED TWICE.  The first rarc_vmem_create reserves one address space for the
// process (RARC_VMEM_SPACE_TIB TiB, default 16; halved until the runtime grants it), slab-aligned by hand; arenas are
// slab-aligned sub-ranges of it handed out first-fit; a destroyed arena returns the part of its range that was never
// backed and RETIRES the part that was.  The space therefore lasts for 16 TiB of slabs mapped over the life of the process
// (a hundred 150 GB indexes); a process that churns through more raises RARC_VMEM_SPACE_TIB.
#include <map>
#include <mutex>
#include <new>
#include <vector>
#include "rarc_common.h"

namespace {
std::mutex g_mu;
char* g_base = nullptr;            // first slab-grid address of the process's reservation
size_t g_bytes = 0, g_slab = 0;    // its size (whole slabs) and THE slab size
std::map<size_t, size_t> g_free;   // free ranges of the space: offset -> length (bytes, whole slabs), coalesced

int space_init(size_t slab) {      // caller holds g_mu
  if (g_base) return RARC_OK;
  size_t tib = 16;
  if (const char* e = getenv("RARC_VMEM_SPACE_TIB")) {
    const long v = atol(e);
    if (v > 0) tib = (size_t)v;
  }
  size_t want = tib << 40;
  void* p = nullptr;
  while (want >= ((size_t)1 << 36)) {          // give up below 64 GiB
    if (hipMemAddressReserve(&p, want + slab, 0, nullptr, 0) == hipSuccess) break;
    (void)hipGetLastError();
    p = nullptr;
    want >>= 1;
  }
  if (!p) {
    rarc_set_error("rarc_vmem_create: hipMemAddressReserve refused every size from %zu TiB down to 64 GiB", tib);
    return RARC_E_HIP;
  }
  g_base = (char*)(((uintptr_t)p + slab - 1) / slab * slab);
  g_bytes = want / slab * slab;
  g_slab = slab;
  g_free.clear();
  g_free[0] = g_bytes;
  return RARC_OK;
}

void space_release(size_t off, size_t len) {   // caller holds g_mu
  auto it = g_free.emplace(off, len).first;
  auto nx = std::next(it);
  if (nx != g_free.end() && it->first + it->second == nx->first) {
    it->second += nx->second;
    g_free.erase(nx);
  }
  if (it != g_free.begin()) {
    auto pv = std::prev(it);
    if (pv->first + pv->second == it->first) {
      pv->second += it->second;
      g_free.erase(it);
    }
  }
}
}  // namespace

struct RarcVmem {
  int device;
  char* base;      // = g_base + offset
  size_t offset;   // of the arena inside the process's address space
  size_t reserved, mapped, slab, gran;
  std::vector<hipMemGenericAllocationHandle_t> slabs;   // one handle per mapped slab, in address order
  std::mutex mu;
};

static hipMemAllocationProp vmem_prop(int device) {
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = device;
  return p;
}

// reserve_bytes: the address range wanted; min_reserve_bytes (0 = reserve_bytes): the least the caller can live with — when
// no free range holds reserve_bytes the largest one that holds min_reserve_bytes is taken whole (an index created "as
// large as the device" does not need the last slab of that; rarc_vmem_reserved says what it got).
extern "C" int rarc_vmem_create(int device, size_t reserve_bytes, size_t min_reserve_bytes, size_t slab_bytes, RarcVmem** out) {
  RARC_REQUIRE(out && reserve_bytes > 0 && device >= 0 && min_reserve_bytes <= reserve_bytes, RARC_E_INVALID,
               "rarc_vmem_create: bad argument");
  *out = nullptr;
  hipMemAllocationProp prop = vmem_prop(device);
  size_t gran = 0;
  RARC_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (gran == 0) gran = 2u << 20;
  if (slab_bytes == 0) slab_bytes = (size_t)RARC_VMEM_DEFAULT_SLAB;
  const size_t slab = (slab_bytes + gran - 1) / gran * gran;
  size_t reserve = (reserve_bytes + slab - 1) / slab * slab;
  const size_t least = min_reserve_bytes ? (min_reserve_bytes + slab - 1) / slab * slab : reserve;
  RarcVmem* v = new (std::nothrow) RarcVmem();
  RARC_REQUIRE(v, RARC_E_INVALID, "rarc_vmem_create: out of host memory");
  {
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = space_init(slab);
    if (rc == RARC_OK && g_slab != slab) {
      rarc_set_error("rarc_vmem_create: this process maps %zu-byte slabs; an arena of %zu-byte slabs cannot coexist with "
                     "them (one piece size per process: see csrc/vmem.hip)", g_slab, slab);
      rc = RARC_E_UNSUPPORTED;
    }
    if (rc == RARC_OK) {
      auto it = g_free.begin();
      while (it != g_free.end() && it->second < reserve) ++it;       // first fit
      if (it == g_free.end()) {                                      // ... else the largest range that holds the minimum
        auto best = g_free.end();
        for (auto j = g_free.begin(); j != g_free.end(); ++j)
          if (j->second >= least && (best == g_free.end() || j->second > best->second)) best = j;
        if (best != g_free.end()) {
          it = best;
          reserve = best->second;
        }
      }
      if (it == g_free.end()) {
        rarc_set_error("rarc_vmem_create: no %zu-byte range left in the process's %zu-byte address space (live arenas hold "
                       "it; RARC_VMEM_SPACE_TIB raises the space, max_rows lowers an index's share)", least, g_bytes);
        rc = RARC_E_WORKSPACE;
      } else {
        const size_t off = it->first, len = it->second;
        g_free.erase(it);
        if (len > reserve) g_free[off + reserve] = len - reserve;
        v->offset = off;
      }
    }
    if (rc != RARC_OK) {
      delete v;
      return rc;
    }
  }
  v->device = device;
  v->base = g_base + v->offset;
  v->reserved = reserve;
  v->mapped = 0;
  v->slab = slab;
  v->gran = gran;
  *out = v;
  return RARC_OK;
}

// Back the arena up to at least min_bytes (rounded up to whole slabs: one physical allocation each).  Never shrinks; on
// failure what was mapped before stays mapped.
extern "C" int rarc_vmem_grow(RarcVmem* v, size_t min_bytes) {
  RARC_REQUIRE(v, RARC_E_INVALID, "rarc_vmem_grow: null arena");
  std::lock_guard<std::mutex> lk(v->mu);
  RARC_REQUIRE(min_bytes <= v->reserved, RARC_E_INVALID, "rarc_vmem_grow: %zu bytes asked of an arena that reserved %zu",
               min_bytes, v->reserved);
  hipMemAllocationProp prop = vmem_prop(v->device);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t target = (min_bytes + v->slab - 1) / v->slab * v->slab;
  while (v->mapped < target) {
    const size_t piece = v->slab;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, piece, &prop, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      rarc_set_error("rarc_vmem_grow: hipMemCreate of %zu bytes failed with %zu bytes mapped: %s", piece, v->mapped,
                     hipGetErrorString(e));
      return RARC_E_HIP;
    }
    e = hipMemMap(v->base + v->mapped, piece, 0, h, 0);
    if (e == hipSuccess) {
      e = hipMemSetAccess(v->base + v->mapped, piece, &acc, 1);
      if (e != hipSuccess) (void)hipMemUnmap(v->base + v->mapped, piece);
    }
    if (e != hipSuccess) {
      (void)hipMemRelease(h);
      (void)hipGetLastError();
      rarc_set_error("rarc_vmem_grow: mapping a slab at offset %zu failed: %s", v->mapped, hipGetErrorString(e));
      return RARC_E_HIP;
    }
    try {
      v->slabs.push_back(h);
    } catch (const std::bad_alloc&) {
      (void)hipMemUnmap(v->base + v->mapped, piece);
      (void)hipMemRelease(h);
      rarc_set_error("rarc_vmem_grow: out of host memory");
      return RARC_E_INVALID;
    }
    v->mapped += piece;
  }
  return RARC_OK;
}

extern "C" void* rarc_vmem_base(const RarcVmem* v) { return v ? (void*)v->base : nullptr; }
extern "C" size_t rarc_vmem_mapped(const RarcVmem* v) { return v ? v->mapped : 0; }
extern "C" size_t rarc_vmem_reserved(const RarcVmem* v) { return v ? v->reserved : 0; }
extern "C" size_t rarc_vmem_slab(const RarcVmem* v) { return v ? v->slab : 0; }
extern "C" size_t rarc_vmem_granularity(const RarcVmem* v) { return v ? v->gran : 0; }

// Unmap and release every slab (the memory goes back to the device at once); of the address range, the part that was
// never backed returns to the process's space, the part that was is retired (see the header).  The caller makes sure no
// kernel still reads the arena.
extern "C" int rarc_vmem_destroy(RarcVmem* v) {
  if (!v) return RARC_OK;
  int rc = RARC_OK;
  for (size_t i = 0; i < v->slabs.size(); ++i) {
    if (hipMemUnmap(v->base + i * v->slab, v->slab) != hipSuccess) rc = RARC_E_HIP;
    if (hipMemRelease(v->slabs[i]) != hipSuccess) rc = RARC_E_HIP;
  }
  if (rc != RARC_OK) {
    (void)hipGetLastError();
    rarc_set_error("rarc_vmem_destroy: releasing the arena failed");
  }
  if (v->reserved > v->mapped) {   // the never-backed tail is clean address space; [offset, offset + mapped) is retired
    std::lock_guard<std::mutex> lk(g_mu);
    space_release(v->offset + v->mapped, v->reserved - v->mapped);
  }
  delete v;
  return rc;
}
