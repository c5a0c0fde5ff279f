// vmem.hip — growable device arenas on HIP virtual memory (see include/rarc.h: rarc_vmem_*).
//
// index.add in the reference appends (faiss grows its std::vector: VectorStore_Faiss.py:199-202).  A row buffer in HBM that
// grows by allocate-bigger-and-copy holds old + new at once — up to 3x the live rows — so a store past a third of the
// part's memory could not grow.  An arena reserves virtual address space for the largest size it may ever reach
// (hipMemAddressReserve: costs no memory) and backs it slab by slab (hipMemCreate + hipMemMap + hipMemSetAccess) as rows
// arrive: the base pointer never moves, nothing is copied, the kernels see one contiguous buffer as before, and the peak
// footprint is the live rows rounded up to a slab.
//
// ONE slab size per process, every piece exactly one slab.  Measured on this runtime (ROCm 7.0.2 / HIP 7.0.51831,
// tools/vmem_probe.py, every sequence in a fresh process): inside one reservation the second mapped piece fixes the size
// of all later ones (2,30 ok; 2,8,8,8 ok then 6 -> hipErrorInvalidValue; 2,2 then 4 -> invalid), and a reservation
// whose address range is reused after hipMemAddressFree by pieces of ANOTHER size maps without error but reads back
// wrong data.  Uniform pieces (any size, power of two or not) behave in every sequence tried, including destroy /
// re-create.  So: the first rarc_vmem_create of a process fixes the slab size; a later call asking for another is refused.
#include <mutex>
#include <new>
#include <vector>
#include "rarc_common.h"

struct RarcVmemSlab {
  hipMemGenericAllocationHandle_t handle;
  size_t bytes;
};
struct RarcVmem {
  int device;
  char* base;          // first slab-grid address inside the reservation [va_base, va_base + va_bytes)
  void* va_base;
  size_t va_bytes;
  size_t reserved, mapped, slab, gran;   // slab: the largest single physical allocation; gran: the mapping granularity
  std::vector<RarcVmemSlab> slabs;
  std::mutex mu;
};

static hipMemAllocationProp vmem_prop(int device) {
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = device;
  return p;
}

extern "C" int rarc_vmem_create(int device, size_t reserve_bytes, size_t slab_bytes, RarcVmem** out) {
  RARC_REQUIRE(out && reserve_bytes > 0 && device >= 0, RARC_E_INVALID, "rarc_vmem_create: bad argument");
  *out = nullptr;
  hipMemAllocationProp prop = vmem_prop(device);
  size_t gran = 0;
  RARC_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (gran == 0) gran = 2u << 20;
  if (slab_bytes == 0) slab_bytes = (size_t)RARC_VMEM_DEFAULT_SLAB;
  size_t slab = (slab_bytes + gran - 1) / gran * gran;
  {
    static std::mutex g_mu;
    static size_t g_slab = 0;        // the process's one piece size (see the header of this file)
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_slab == 0) g_slab = slab;
    RARC_REQUIRE(g_slab == slab, RARC_E_UNSUPPORTED,
                 "rarc_vmem_create: this process maps %zu-byte slabs; an arena of %zu-byte slabs cannot coexist with them "
                 "(HIP runtime: pieces of different sizes in reused address ranges corrupt mappings)", g_slab, slab);
  }
  size_t reserve = (reserve_bytes + slab - 1) / slab * slab;
  void* base = nullptr;
  // Reservations start on the slab grid (address % slab == 0): every piece any arena of this process ever maps then
  // covers one of the SAME address ranges [k * slab, (k + 1) * slab) — an address range that is freed and reserved again
  // (another index, another size) is re-mapped with pieces identical to the ones the runtime has seen there (pieces at
  // other phases of a reused range read back wrong data on this runtime: tests in sequence showed it, see the header).
  // Slab-aligned addresses also let the driver use large page-table fragments: the scan streams an arena at HBM rate.
  // (the runtime ignores hipMemAddressReserve's alignment argument — measured: it hands out 2 MiB-aligned ranges — so one
  // extra slab is reserved and the arena starts at the first grid point inside the range)
  RARC_HIP_CHECK(hipMemAddressReserve(&base, reserve + slab, slab, nullptr, 0));
  void* const va_base = base;
  const size_t va_bytes = reserve + slab;
  base = (void*)(((uintptr_t)base + slab - 1) / slab * slab);
  RarcVmem* v = new (std::nothrow) RarcVmem();
  if (!v) {
    (void)hipMemAddressFree(va_base, va_bytes);
    rarc_set_error("rarc_vmem_create: out of host memory");
    return RARC_E_INVALID;
  }
  v->device = device;
  v->base = (char*)base;
  v->va_base = va_base;
  v->va_bytes = va_bytes;
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
      v->slabs.push_back(RarcVmemSlab{h, piece});
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

// Unmap and release every slab, free the address range.  The caller makes sure no kernel still reads the arena.
extern "C" int rarc_vmem_destroy(RarcVmem* v) {
  if (!v) return RARC_OK;
  int rc = RARC_OK;
  size_t at = 0;
  for (const RarcVmemSlab& sl : v->slabs) {
    if (hipMemUnmap(v->base + at, sl.bytes) != hipSuccess) rc = RARC_E_HIP;
    if (hipMemRelease(sl.handle) != hipSuccess) rc = RARC_E_HIP;
    at += sl.bytes;
  }
  if (hipMemAddressFree(v->va_base, v->va_bytes) != hipSuccess) rc = RARC_E_HIP;
  if (rc != RARC_OK) {
    (void)hipGetLastError();
    rarc_set_error("rarc_vmem_destroy: releasing the arena failed");
  }
  delete v;
  return rc;
}
