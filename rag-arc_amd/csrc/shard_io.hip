// shard_io.hip — bulk movement of corpus shards between a file and HBM (C-ABI: rarc_file_to_device, rarc_device_to_file).
//
// Counterpart of the native I/O behind the reference's persistence calls — faiss.write_index / faiss.read_index at
// encapsulation/database/vector_db/VectorStore_Faiss.py:438 and :467 — for an index whose rows live in HBM.
// A 100M x 768 fp16 shard is 153.6 GB: it must never exist as a host array.  Both directions therefore stream through a
// small ring of PINNED host slots owned by the caller (hipHostMalloc'ed memory, e.g. a torch pin_memory tensor):
//
//   file -> device   T worker threads; each owns two slots and alternates:  pread(chunk -> slot A) | hipMemcpyAsync(A -> HBM)
//                    | pread(next chunk -> slot B) while A is in flight | ...
//   device -> file   the mirror image: hipMemcpyAsync(HBM -> slot A) | pwrite(slot B, the previous chunk) | ...
//
// Chunks are handed out by one atomic counter, so the threads need no other coordination, and the host footprint is the
// ring (2·T slots) whatever the shard size.  With RARC_IO_DIRECT the file is opened O_DIRECT as well and every chunk whose
// offset / length / slot are 4096-aligned bypasses the page cache (the DMA of the storage device lands in the pinned slot
// the GPU's DMA then reads: no copy by a CPU); unaligned pieces (the tail of a section) go through the buffered descriptor.
// A file system that refuses O_DIRECT (tmpfs) is served buffered, reported in RarcIoStats.direct = 0.
//
// The calls are synchronous (the data is in place when they return); the DMA is enqueued on the caller's `stream`, hence
// ordered behind whatever the caller queued there before.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "rarc_common.h"

namespace {

struct Chunk {
  int64_t file_off, bytes, dev_off;
};

struct IoJob {
  bool to_device;
  int fd_buf = -1, fd_direct = -1;
  char* d_base = nullptr;
  char* staging = nullptr;
  size_t slot_bytes = 0;
  int device = 0;
  hipStream_t caller_stream = nullptr;
  std::vector<Chunk> chunks;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::atomic<int> io_failed{0};   // the failure was a file operation (reported as RARC_E_IO, not RARC_E_HIP)
  std::atomic<int64_t> direct_bytes{0};
  std::atomic<int64_t> file_ns{0}, copy_wait_ns{0};
  char err[256] = "";
  std::atomic_flag err_lock = ATOMIC_FLAG_INIT;

  void fail(const char* what, const char* detail) {
    if (!err_lock.test_and_set()) snprintf(err, sizeof(err), "%s: %s", what, detail);
    failed.store(1);
  }
};

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// whole-chunk pread / pwrite (short transfers are continued; EINTR retried)
bool xfer_file(IoJob& job, bool direct_ok, bool write, char* buf, int64_t bytes, int64_t off) {
  const bool aligned = direct_ok && job.fd_direct >= 0 && (off % 4096 == 0) && (bytes % 4096 == 0) && ((uintptr_t)buf % 4096 == 0);
  const int fd = aligned ? job.fd_direct : job.fd_buf;
  int64_t done = 0;
  while (done < bytes) {
    const ssize_t r = write ? pwrite(fd, buf + done, (size_t)(bytes - done), (off_t)(off + done))
                            : pread(fd, buf + done, (size_t)(bytes - done), (off_t)(off + done));
    if (r < 0) {
      if (errno == EINTR) continue;
      job.io_failed.store(1);
      job.fail(write ? "pwrite" : "pread", strerror(errno));
      return false;
    }
    if (r == 0) {
      job.io_failed.store(1);
      job.fail(write ? "pwrite" : "pread", "unexpected end of file");
      return false;
    }
    done += r;
    if (aligned && done < bytes && (done % 4096)) {  // a short direct transfer left us unaligned: finish buffered
      return xfer_file(job, false, write, buf + done, bytes - done, off + done);
    }
  }
  if (aligned) job.direct_bytes += bytes;
  return true;
}

void worker(IoJob* jp, int w) {
  IoJob& job = *jp;
  if (hipSetDevice(job.device) != hipSuccess) { job.fail("hipSetDevice", "worker thread"); return; }
  // The DMA goes onto the CALLER's stream (HIP streams may be fed from several threads).  Measured on MI355X / ROCm 7.2:
  // every additional HIP stream costs the process 90-190 MB of host memory for good (one stream per worker: +730 MB on
  // the first 8-thread call), and buys nothing — one stream carries the PCIe link's rate (tools/lab/persist_rss_probe.py).
  const hipStream_t st = job.caller_stream;
  hipEvent_t ev[2] = {nullptr, nullptr};
  if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
    job.fail("hipEventCreate", "worker thread");
    return;
  }
  char* slot[2] = {job.staging + (size_t)(2 * w) * job.slot_bytes, job.staging + (size_t)(2 * w + 1) * job.slot_bytes};
  bool busy[2] = {false, false};   // a copy involving the slot has been enqueued and its event recorded
  Chunk held[2] = {};              // device -> file: the chunk sitting in the slot, still to be written
  int p = 0;
  auto sync_slot = [&](int s) -> bool {
    if (!busy[s]) return true;
    const int64_t t0 = now_ns();
    const hipError_t e = hipEventSynchronize(ev[s]);
    job.copy_wait_ns += now_ns() - t0;
    busy[s] = false;
    if (e != hipSuccess) { job.fail("hipEventSynchronize", hipGetErrorString(e)); return false; }
    return true;
  };
  auto flush_slot = [&](int s) -> bool {   // device -> file: wait for the slot's copy, write it out
    if (!busy[s]) return true;
    if (!sync_slot(s)) return false;
    const int64_t t0 = now_ns();
    const bool ok = xfer_file(job, true, true, slot[s], held[s].bytes, held[s].file_off);
    job.file_ns += now_ns() - t0;
    return ok;
  };
  while (!job.failed.load()) {
    const size_t i = job.next.fetch_add(1);
    if (i >= job.chunks.size()) break;
    const Chunk c = job.chunks[i];
    if (job.to_device) {
      if (!sync_slot(p)) break;                       // the slot's previous upload has left it
      const int64_t t0 = now_ns();
      const bool ok = xfer_file(job, true, false, slot[p], c.bytes, c.file_off);
      job.file_ns += now_ns() - t0;
      if (!ok) break;
      hipError_t e = hipMemcpyAsync(job.d_base + c.dev_off, slot[p], (size_t)c.bytes, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipEventRecord(ev[p], st);
      if (e != hipSuccess) { job.fail("hipMemcpyAsync (H2D)", hipGetErrorString(e)); break; }
      busy[p] = true;
    } else {
      if (!flush_slot(p)) break;                      // the chunk that sat in this slot is on disk
      hipError_t e = hipMemcpyAsync(slot[p], job.d_base + c.dev_off, (size_t)c.bytes, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(ev[p], st);
      if (e != hipSuccess) { job.fail("hipMemcpyAsync (D2H)", hipGetErrorString(e)); break; }
      busy[p] = true;
      held[p] = c;
      if (!flush_slot(p ^ 1)) break;                  // write the previous chunk while this one is in flight
    }
    p ^= 1;
  }
  if (job.to_device) { (void)sync_slot(0); (void)sync_slot(1); }
  else if (!job.failed.load()) { (void)flush_slot(p); (void)flush_slot(p ^ 1); }   // older chunk first
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
}

int run(bool to_device, const char* path, int n_seg, const int64_t* file_off, const int64_t* bytes, const int64_t* dev_off,
        void* d_base, int64_t d_capacity, void* h_staging, size_t staging_bytes, int n_threads, int flags, void* stream,
        RarcIoStats* stats) {
  const char* fn = to_device ? "rarc_file_to_device" : "rarc_device_to_file";
  RARC_REQUIRE(path && n_seg >= 0 && (n_seg == 0 || (file_off && bytes && dev_off)), RARC_E_INVALID, "%s: null argument", fn);
  RARC_REQUIRE(d_base && h_staging, RARC_E_INVALID, "%s: null buffer", fn);
  RARC_REQUIRE(n_threads >= 1 && n_threads <= 64, RARC_E_INVALID, "%s: n_threads must be 1..64", fn);
  RARC_REQUIRE((uintptr_t)h_staging % 4096 == 0, RARC_E_INVALID, "%s: the staging buffer must be 4096-byte aligned", fn);
  const size_t slot_bytes = (staging_bytes / (size_t)(2 * n_threads)) & ~(size_t)4095;
  RARC_REQUIRE(slot_bytes >= 65536, RARC_E_WORKSPACE, "%s: staging buffer too small: %zu bytes for %d threads (two slots of >= 64 KiB each)",
               fn, staging_bytes, n_threads);
  hipPointerAttribute_t attr;
  RARC_REQUIRE(hipPointerGetAttributes(&attr, h_staging) == hipSuccess && attr.type == hipMemoryTypeHost, RARC_E_INVALID,
               "%s: the staging buffer must be pinned host memory (hipHostMalloc / torch pin_memory)", fn);
  IoJob job;
  job.to_device = to_device;
  int64_t total = 0;
  try {
    size_t n_chunks = 0;
    for (int s = 0; s < n_seg; ++s) n_chunks += bytes[s] > 0 ? (size_t)((bytes[s] + (int64_t)slot_bytes - 1) / (int64_t)slot_bytes) : 0;
    job.chunks.reserve(n_chunks);
  } catch (const std::exception&) {
    rarc_set_error("%s: out of host memory for the chunk list", fn);
    return RARC_E_INVALID;
  }
  for (int s = 0; s < n_seg; ++s) {
    RARC_REQUIRE(file_off[s] >= 0 && bytes[s] >= 0 && dev_off[s] >= 0 && dev_off[s] + bytes[s] <= d_capacity, RARC_E_INVALID,
                 "%s: segment %d (file %lld, %lld bytes, device offset %lld) outside the device buffer of %lld bytes", fn, s,
                 (long long)file_off[s], (long long)bytes[s], (long long)dev_off[s], (long long)d_capacity);
    for (int64_t o = 0; o < bytes[s]; o += (int64_t)slot_bytes) {
      const int64_t len = bytes[s] - o < (int64_t)slot_bytes ? bytes[s] - o : (int64_t)slot_bytes;
      job.chunks.push_back({file_off[s] + o, len, dev_off[s] + o});
    }
    total += bytes[s];
  }
  const int oflags = to_device ? O_RDONLY : (O_WRONLY | O_CREAT);
  job.fd_buf = open(path, oflags | O_CLOEXEC, 0644);
  RARC_REQUIRE(job.fd_buf >= 0, RARC_E_IO, "%s: cannot open %s: %s", fn, path, strerror(errno));
  if (flags & RARC_IO_DIRECT) job.fd_direct = open(path, oflags | O_CLOEXEC | O_DIRECT, 0644);   // may fail (tmpfs): buffered then
  int64_t need = 0;      // where the furthest segment ends in the file
  for (int s = 0; s < n_seg; ++s) need = file_off[s] + bytes[s] > need ? file_off[s] + bytes[s] : need;
  if (to_device) {
    struct stat sb = {};
    const int st_rc = fstat(job.fd_buf, &sb);
    if (st_rc != 0 || (int64_t)sb.st_size < need) {
      const int err_no = errno;
      close(job.fd_buf);
      if (job.fd_direct >= 0) close(job.fd_direct);
      if (st_rc != 0) rarc_set_error("%s: fstat(%s): %s", fn, path, strerror(err_no));
      else rarc_set_error("%s: %s is shorter (%lld bytes) than the segments ask for (%lld)", fn, path, (long long)sb.st_size, (long long)need);
      return st_rc != 0 ? RARC_E_IO : RARC_E_INVALID;
    }
  } else if ((flags & RARC_IO_TRUNCATE) && ftruncate(job.fd_buf, (off_t)need) != 0) {
    const int err_no = errno;
    close(job.fd_buf);
    if (job.fd_direct >= 0) close(job.fd_direct);
    rarc_set_error("%s: ftruncate(%s, %lld): %s", fn, path, (long long)need, strerror(err_no));
    return RARC_E_IO;
  }
  job.d_base = (char*)d_base;
  job.staging = (char*)h_staging;
  job.slot_bytes = slot_bytes;
  int rc = RARC_OK;
  hipError_t e = hipGetDevice(&job.device);
  job.caller_stream = (hipStream_t)stream;
  const int64_t t0 = now_ns();
  if (e == hipSuccess) {
    const int n_workers = (int)job.chunks.size() < n_threads ? (job.chunks.empty() ? 0 : (int)job.chunks.size()) : n_threads;
    std::vector<std::thread> th;
    try {               // (thread creation can fail — EAGAIN under a process limit — and must not unwind through extern "C")
      th.reserve((size_t)n_workers);
      for (int w = 0; w < n_workers; ++w) th.emplace_back(worker, &job, w);
    } catch (const std::exception& ex) {
      job.fail("starting a worker thread", ex.what());     // the workers already running see `failed` and stop
    }
    for (auto& t : th) t.join();
    if (!to_device && !job.failed.load() && (flags & RARC_IO_FSYNC) && fsync(job.fd_buf) != 0) {
      job.io_failed.store(1);
      job.fail("fsync", strerror(errno));
    }
  } else {
    job.fail("hipGetDevice", hipGetErrorString(e));
  }
  if (e == hipSuccess && hipStreamSynchronize(job.caller_stream) != hipSuccess) job.fail("hipStreamSynchronize", "caller's stream");
  const int64_t t1 = now_ns();
  if (job.fd_direct >= 0) close(job.fd_direct);
  if (close(job.fd_buf) != 0 && !to_device && !job.failed.load()) {
    job.io_failed.store(1);
    job.fail("close", strerror(errno));
  }
  if (job.failed.load()) {
    rarc_set_error("%s(%s): %s", fn, path, job.err);
    rc = job.io_failed.load() ? RARC_E_IO : RARC_E_HIP;
  }
  if (stats) {
    stats->bytes = total;
    stats->seconds = (double)(t1 - t0) * 1e-9;
    stats->file_seconds = (double)job.file_ns.load() * 1e-9;        // summed over the workers
    stats->copy_wait_seconds = (double)job.copy_wait_ns.load() * 1e-9;
    stats->direct_bytes = job.direct_bytes.load();
    stats->n_chunks = (int64_t)job.chunks.size();
    stats->slot_bytes = (int64_t)slot_bytes;
    stats->n_threads = n_threads;
    stats->direct = job.fd_direct >= 0 ? 1 : 0;
  }
  return rc;
}

}  // namespace

extern "C" int rarc_file_to_device(const char* path, int n_seg, const int64_t* h_file_off, const int64_t* h_bytes,
                                   const int64_t* h_dev_off, void* d_base, int64_t d_capacity_bytes, void* h_staging,
                                   size_t staging_bytes, int n_threads, int flags, void* stream, RarcIoStats* stats) {
  RARC_RANGE();
  return run(true, path, n_seg, h_file_off, h_bytes, h_dev_off, d_base, d_capacity_bytes, h_staging, staging_bytes, n_threads,
             flags, stream, stats);
}

extern "C" int rarc_device_to_file(const char* path, int n_seg, const int64_t* h_file_off, const int64_t* h_bytes,
                                   const int64_t* h_dev_off, const void* d_base, int64_t d_capacity_bytes, void* h_staging,
                                   size_t staging_bytes, int n_threads, int flags, void* stream, RarcIoStats* stats) {
  RARC_RANGE();
  return run(false, path, n_seg, h_file_off, h_bytes, h_dev_off, const_cast<void*>(d_base), d_capacity_bytes, h_staging,
             staging_bytes, n_threads, flags, stream, stats);
}
