// host_feed_probe.hip -- what does it cost the HOST to get packed rows from a file into HBM?  (round 5, VERDICT r04 "next" 1c)
// Measures, for one file of S bytes in the page cache, wall time and process CPU time (user + sys, all threads: getrusage) of
//   A  mmap + memcpy into hipHostMalloc'ed staging + hipMemcpyAsync        (round 4's path: api.hip parallel_copy, one thread here)
//   B  pread() into the pinned staging + hipMemcpyAsync                    (kernel copy instead of page faults + user copy)
//   C  mmap + hipHostRegister(window) + hipMemcpyAsync + hipHostUnregister (the device reads the page cache by DMA)
//   C2 as C with hipHostRegisterReadOnly
//   D  O_DIRECT pread() into the pinned staging + hipMemcpyAsync           (storage DMA; bounded by the disk)
//   E  hipMemcpyAsync from a resident pinned buffer                        (the PCIe rate itself)
// Build: hipcc -O2 --offload-arch=gfx950 host_feed_probe.hip -o host_feed_probe     Run: ./host_feed_probe [dir] [GB] [window MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/statfs.h>

static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu()
{
    rusage r; getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); ok = false; } } while (0)

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t S = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (1u << 30));
    const size_t W = (size_t)(argc > 3 ? atoi(argv[3]) : 160) << 20;
    const std::string path = dir + "/host_feed_probe.bin";
    struct statfs sf; if (statfs(dir.c_str(), &sf) == 0) printf("fs of %s: type 0x%lx, block %ld\n", dir.c_str(), (long)sf.f_type, (long)sf.f_bsize);
    {
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fd < 0) { perror("create"); return 1; }
        std::string buf(1 << 24, 'x'); for (size_t i = 0; i < buf.size(); i += 7) buf[i] = (char)(i * 2654435761u >> 13);
        for (size_t o = 0; o < S; o += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) { perror("write"); return 1; }
        fsync(fd); close(fd);
    }
    bool ok = true;
    if (getenv("PROBE_BLOCKING_DEVICE")) { CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync)); printf("device flag hipDeviceScheduleBlockingSync set\n"); }
    uint8_t *d = nullptr, *pin[2] = {nullptr, nullptr};
    CK(hipSetDevice(0)); CK(hipMalloc((void **)&d, W)); CK(hipHostMalloc((void **)&pin[0], W, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin[1], W, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    memset(pin[0], 1, W); memset(pin[1], 2, W);
    auto report = [&](const char *name, double w, double c, bool good) {
        printf("%-3s %s: %.3f s wall = %.2f GB/s, %.3f CPU-s = %.1f ns CPU per 632-byte row (%.4f CPU-s per million rows)\n", name, good ? "ok  " : "FAIL",
               w, S / w / 1e9, c, c / (S / 632.0) * 1e9, c / (S / 632.0) * 1e6);
        fflush(stdout);
    };
    // F: what a WAIT costs the waiting thread: 24 x 160 MB of copies queued, then one wait (CPU-s ~ wall = the wait spins)
    for (int mode = 0; mode < 3; ++mode) {
      ok = true; hipEvent_t e; CK(hipEventCreateWithFlags(&e, mode == 2 ? (hipEventBlockingSync | hipEventDisableTiming) : hipEventDisableTiming));
      for (int k = 0; k < 24; ++k) CK(hipMemcpyAsync(d, pin[k & 1], W, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(e, st));
      const double w0 = wall(), c0 = cpu();
      if (mode == 0) CK(hipStreamSynchronize(st)); else CK(hipEventSynchronize(e));
      printf("F%d  %s: waited %.4f s wall, %.4f CPU-s (%s)\n", mode, ok ? "ok  " : "FAIL", wall() - w0, cpu() - c0,
             mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventSynchronize, default event" : "hipEventSynchronize, hipEventBlockingSync event");
      CK(hipEventDestroy(e)); }
    // E: PCIe alone
    { ok = true; const double w0 = wall(), c0 = cpu();
      for (size_t o = 0, k = 0; o < S; o += W, ++k) { CK(hipMemcpyAsync(d, pin[k & 1], W, hipMemcpyHostToDevice, st)); }
      CK(hipStreamSynchronize(st)); report("E", wall() - w0, cpu() - c0, ok); }
    for (int rep = 0; rep < 2; ++rep) {
    // A: mmap + memcpy
    { ok = true; int fd = open(path.c_str(), O_RDONLY); void *m = mmap(nullptr, S, PROT_READ, MAP_SHARED, fd, 0); madvise(m, S, MADV_SEQUENTIAL);
      const double w0 = wall(), c0 = cpu();
      for (size_t o = 0, k = 0; o < S; o += W, ++k) { const size_t n = std::min(W, S - o); CK(hipEventSynchronize(ev[k & 1])); memcpy(pin[k & 1], (uint8_t *)m + o, n);
          CK(hipMemcpyAsync(d, pin[k & 1], n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k & 1], st)); }
      CK(hipStreamSynchronize(st)); report("A", wall() - w0, cpu() - c0, ok); munmap(m, S); close(fd); }
    // B: pread into pinned
    { ok = true; int fd = open(path.c_str(), O_RDONLY);
      const double w0 = wall(), c0 = cpu();
      for (size_t o = 0, k = 0; o < S; o += W, ++k) { const size_t n = std::min(W, S - o); CK(hipEventSynchronize(ev[k & 1]));
          for (size_t g = 0; g < n;) { ssize_t r = pread(fd, pin[k & 1] + g, n - g, o + g); if (r <= 0) { ok = false; break; } g += r; }
          CK(hipMemcpyAsync(d, pin[k & 1], n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k & 1], st)); }
      CK(hipStreamSynchronize(st)); report("B", wall() - w0, cpu() - c0, ok); close(fd); }
    }
    // C: register windows of the mapping
    for (int ro = 0; ro < 3; ++ro) {
      ok = true; int fd = open(path.c_str(), O_RDONLY);
      void *m = mmap(nullptr, S, ro == 2 ? (PROT_READ | PROT_WRITE) : PROT_READ, ro == 2 ? MAP_PRIVATE : MAP_SHARED, fd, 0);
      const double w0 = wall(), c0 = cpu(); double treg = 0, tunreg = 0;
      for (size_t o = 0; o < S && ok; o += W) { const size_t n = std::min(W, S - o);
          const double t0 = wall();
          CK(hipHostRegister((uint8_t *)m + o, n, ro == 1 ? hipHostRegisterReadOnly : hipHostRegisterDefault)); if (!ok) break;
          const double t1 = wall(); treg += t1 - t0;
          CK(hipMemcpyAsync(d, (uint8_t *)m + o, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
          const double t2 = wall();
          CK(hipHostUnregister((uint8_t *)m + o)); tunreg += wall() - t2; }
      const char *nm = ro == 0 ? "C" : ro == 1 ? "C2" : "C3";
      report(nm, wall() - w0, cpu() - c0, ok);
      printf("    (%s: %s; register %.3f s, unregister %.3f s of the wall time; copies synchronous here)\n", nm,
             ro == 0 ? "MAP_SHARED read-only mapping, default flags" : ro == 1 ? "MAP_SHARED read-only mapping, hipHostRegisterReadOnly" : "MAP_PRIVATE writable mapping, default flags", treg, tunreg);
      munmap(m, S); close(fd); }
    // G: as the command line does it (pyseer_amd/input.py): windows at odd offsets of ONE mapping, registered one ahead of the copy, unregistered after it
    for (int var = 0; var < 4; ++var) {
      ok = true; int fd = open(path.c_str(), O_RDONLY);
      void *m = mmap(nullptr, S, PROT_READ, MAP_SHARED, fd, 0);
      if (var & 1) madvise(m, S, MADV_SEQUENTIAL);
      const size_t skew = (var & 2) ? 1234567 : 0, pg = 4096;
      const double w0 = wall(), c0 = cpu(); double treg = 0, tunreg = 0, tcopy = 0;
      auto win = [&](size_t o, uintptr_t &lo, size_t &len) { const size_t n = std::min(W, S - o); const uintptr_t a = (uintptr_t)m + o + (o ? skew : 0), b = std::min((uintptr_t)m + S, a + n);
                                                               lo = a & ~(pg - 1); len = ((b + pg - 1) & ~(pg - 1)) - lo; if (lo + len > (uintptr_t)m + S) len = (uintptr_t)m + S - lo; };
      uintptr_t lo0; size_t len0; win(0, lo0, len0);
      { const double t0 = wall(); CK(hipHostRegister((void *)lo0, len0, hipHostRegisterDefault)); treg += wall() - t0; }
      for (size_t o = 0; o < S && ok; o += W) {
          uintptr_t lo, lo1 = 0; size_t len, len1 = 0; win(o, lo, len);
          const double t1 = wall();
          CK(hipMemcpyAsync(d, (void *)lo, std::min(len, W), hipMemcpyHostToDevice, st));
          if (o + W < S) { win(o + W, lo1, len1); if (lo1 < lo + len) { len1 -= (lo + len - lo1); lo1 = lo + len; }      // (windows must not overlap)
                           const double t0 = wall(); CK(hipHostRegister((void *)lo1, len1, hipHostRegisterDefault)); treg += wall() - t0; }
          CK(hipStreamSynchronize(st));
          const double t2 = wall(); tcopy += t2 - t1;
          CK(hipHostUnregister((void *)lo)); tunreg += wall() - t2; }
      char nm[8]; snprintf(nm, sizeof nm, "G%d", var);
      report(nm, wall() - w0, cpu() - c0, ok);
      printf("    (%s: madvise(SEQUENTIAL) %s, windows %s; register %.3f s (%.1f ms per window), unregister %.3f s, copy + overlapped register %.3f s)\n", nm, (var & 1) ? "yes" : "no",
             (var & 2) ? "at odd offsets" : "page-aligned", treg, treg / ((S + W - 1) / W) * 1e3, tunreg, tcopy);
      munmap(m, S); close(fd); }
    // H: the whole mapping registered once
    { ok = true; int fd = open(path.c_str(), O_RDONLY); void *m = mmap(nullptr, S, PROT_READ, MAP_SHARED, fd, 0);
      const double w0 = wall(), c0 = cpu();
      const double t0 = wall(); CK(hipHostRegister(m, S, hipHostRegisterDefault)); const double treg = wall() - t0;
      for (size_t o = 0; o < S && ok; o += W) CK(hipMemcpyAsync(d, (uint8_t *)m + o, std::min(W, S - o), hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      const double t2 = wall(); CK(hipHostUnregister(m)); const double tun = wall() - t2;
      report("H", wall() - w0, cpu() - c0, ok);
      printf("    (H: one registration of the whole mapping: register %.3f s, unregister %.3f s)\n", treg, tun);
      munmap(m, S); close(fd); }
    // D: O_DIRECT
    { ok = true; int fd = open(path.c_str(), O_RDONLY | O_DIRECT);
      if (fd < 0) { printf("D   O_DIRECT open failed: %s\n", strerror(errno)); }
      else { const double w0 = wall(), c0 = cpu();
        for (size_t o = 0, k = 0; o < S; o += W, ++k) { const size_t n = std::min(W, S - o); CK(hipEventSynchronize(ev[k & 1]));
            for (size_t g = 0; g < n;) { ssize_t r = pread(fd, pin[k & 1] + g, n - g, o + g); if (r <= 0) { printf("    O_DIRECT pread: %s\n", strerror(errno)); ok = false; break; } g += r; }
            if (!ok) break;
            CK(hipMemcpyAsync(d, pin[k & 1], n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k & 1], st)); }
        CK(hipStreamSynchronize(st)); report("D", wall() - w0, cpu() - c0, ok); close(fd); } }
    unlink(path.c_str());
    return 0;
}
