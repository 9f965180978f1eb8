// rt_multi.hip — one render call across several GPUs of one node, behind the C ABI (include/rt_abi.h, rt_multi_*).
//
// The reference's seam is one blocking call, Camera::render (src/camera/mod.rs:54-58) reached through Scene::render
// (src/scene.rs:30-32); its pixels are plain sums over independent samples (mod.rs:151-172).  rt_multi keeps that shape
// for a Rust / C caller with N GPUs: one rt_ctx and one persistent host thread per GPU, GPU i renders the residue
// class i (mod N) of the per-pixel sample indices for the WHOLE frame (perfect load balance: every GPU sees every
// pixel; the counter-based RNG makes a path a function of its sample index alone), and the partial f32 frames are
// summed on the first device of the list by ONE kernel that reads the other GPUs' frames in place over xGMI.
//
// Why a peer-reading sum kernel and not a ring collective: xGMI is point-to-point — the root has its own link to each
// of the other seven GPUs, so seven 9.7 MB frames (C2) arrive concurrently at ~150 GB/s each (~65 us), with one kernel
// launch and no staging copies; a ring reduce pushes the whole frame through every link in turn and pays a hop latency
// per step.  (bench.py, one process per GPU, uses RCCL's reduce through torch.distributed instead: both are covered.)
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_internal.h"

#define RT_MULTI_MAX_DEVICES 16

struct SumParts {
  const float* p[RT_MULTI_MAX_DEVICES];
};
// out[i] = parts[0][i] + parts[1][i] + ... in list order (deterministic); peers' frames are read over xGMI
__global__ __launch_bounds__(256) void k_sum_frames(SumParts parts, int n, float* __restrict__ out, size_t count) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = count / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(parts.p[0])[i];
    for (int j = 1; j < n; j++) {
      const float4 b = reinterpret_cast<const float4*>(parts.p[j])[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    float a = parts.p[0][i];
    for (int j = 1; j < n; j++) a += parts.p[j][i];
    out[i] = a;
  }
}

namespace {

struct Job {
  const rt_camera* cam = nullptr;
  uint64_t seed = 0;
  uint32_t begin = 0, end = 0, stride = 1;
};

struct Worker {
  int device = 0;
  rt_ctx* ctx = nullptr;
  rt_scene* scene = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;     // recorded on `stream` after this device's render
  float* d_part = nullptr;       // this device's partial frame
  size_t part_cap = 0;           // floats
  std::thread thread;
  int rc = RT_OK;                // result of the last job
  std::string err;
};

}  // namespace

struct rt_multi {
  std::vector<Worker> w;
  std::string err;
  // job hand-off to the persistent workers
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  uint64_t generation = 0;  // bumped for every job
  int pending = 0;
  bool quit = false;
  enum Kind { RENDER, UPLOAD } kind = RENDER;
  const Job* job_table = nullptr;        // RENDER: one entry per device (its residue class of the sample indices)
  const rt_scene_desc* desc = nullptr;   // UPLOAD
  float* d_total = nullptr;  // on the root device: the summed frame
  size_t total_cap = 0;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr;  // root stream: before the reduce / after it / after the copy
  rt_multi_stats stats{};
  bool stats_valid = false;
};

static std::string g_multi_init_err;

static int mfail(rt_multi* m, int code, const std::string& msg) {
  if (m) m->err = msg;
  else g_multi_init_err = msg;
  return code;
}

static void run_job(rt_multi* m, Worker& w, int index) {
  w.rc = RT_OK;
  w.err.clear();
  if (hipSetDevice(w.device) != hipSuccess) { w.rc = RT_E_HIP; w.err = "hipSetDevice failed"; return; }
  if (m->kind == rt_multi::UPLOAD) {
    if (w.scene) { rt_scene_free(w.ctx, w.scene); w.scene = nullptr; }
    w.rc = rt_scene_upload(w.ctx, m->desc, &w.scene);
    if (w.rc != RT_OK) w.err = rt_last_error(w.ctx);
    return;
  }
  const Job& j = m->job_table[index];
  const size_t n_val = (size_t)j.cam->width * j.cam->height * 3;
  if (w.part_cap < n_val) {
    if (w.d_part) (void)rti::dev_free(w.d_part);
    w.d_part = nullptr; w.part_cap = 0;
    if (rti::dev_malloc(&w.d_part, n_val * sizeof(float)) != hipSuccess) { w.rc = RT_E_OOM; w.err = "hipMalloc of the partial frame failed"; return; }
    w.part_cap = n_val;
  }
  w.rc = rt_render_device(w.ctx, w.scene, j.cam, j.seed, j.begin, j.end, j.stride, w.d_part, w.stream);
  if (w.rc != RT_OK) { w.err = rt_last_error(w.ctx); return; }
  if (hipEventRecord(w.done, w.stream) != hipSuccess) { w.rc = RT_E_HIP; w.err = "hipEventRecord failed"; }
}

static void worker_main(rt_multi* m, int index) {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(m->mu);
      m->cv_job.wait(lk, [&] { return m->quit || m->generation != seen; });
      if (m->quit) return;
      seen = m->generation;
    }
    run_job(m, m->w[(size_t)index], index);
    {
      std::lock_guard<std::mutex> lk(m->mu);
      if (--m->pending == 0) m->cv_done.notify_all();
    }
  }
}

// hands the current job to all workers and waits until every one has ISSUED its work (GPU work itself is asynchronous)
static int dispatch(rt_multi* m) {
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->pending = (int)m->w.size();
    m->generation++;
  }
  m->cv_job.notify_all();
  {
    std::unique_lock<std::mutex> lk(m->mu);
    m->cv_done.wait(lk, [&] { return m->pending == 0; });
  }
  for (size_t i = 0; i < m->w.size(); i++)
    if (m->w[i].rc != RT_OK) return mfail(m, m->w[i].rc, "device " + std::to_string(m->w[i].device) + ": " + m->w[i].err);
  return RT_OK;
}

extern "C" {

const char* rt_multi_last_error(rt_multi* m) { return m ? m->err.c_str() : g_multi_init_err.c_str(); }

int rt_multi_device_count(rt_multi* m) { return m ? (int)m->w.size() : 0; }

int rt_multi_shutdown(rt_multi* m) {
  if (!m) return RT_OK;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->quit = true;
  }
  m->cv_job.notify_all();
  for (Worker& w : m->w)
    if (w.thread.joinable()) w.thread.join();
  for (Worker& w : m->w) {
    if (!w.ctx) continue;  // rt_init failed for this entry (or never ran): nothing of it exists
    (void)hipSetDevice(w.device);
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    if (w.scene) rt_scene_free(w.ctx, w.scene);
    if (w.d_part) (void)rti::dev_free(w.d_part);
    if (w.done) (void)rti::event_destroy(w.done);
    if (w.stream) (void)rti::stream_destroy(w.stream);
    if (w.ctx) rt_shutdown(w.ctx);
  }
  if (!m->w.empty() && m->w[0].ctx) (void)hipSetDevice(m->w[0].device);
  if (m->d_total) (void)rti::dev_free(m->d_total);
  if (m->ev_t0) (void)rti::event_destroy(m->ev_t0);
  if (m->ev_t1) (void)rti::event_destroy(m->ev_t1);
  if (m->ev_t2) (void)rti::event_destroy(m->ev_t2);
  delete m;
  return RT_OK;
}

int rt_multi_init(const int* device_ids, int n_devices, rt_multi** out) {
  if (!out) return mfail(nullptr, RT_E_INVALID_ARG, "rt_multi_init: out is NULL");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > RT_MULTI_MAX_DEVICES)
    return mfail(nullptr, RT_E_INVALID_ARG, "rt_multi_init: need 1.." + std::to_string(RT_MULTI_MAX_DEVICES) + " device ids");
  rt_multi* m = new (std::nothrow) rt_multi();
  if (!m) return mfail(nullptr, RT_E_OOM, "rt_multi_init: out of host memory");
  try {
    m->w.resize((size_t)n_devices);
  } catch (...) {
    delete m;
    return mfail(nullptr, RT_E_OOM, "rt_multi_init: out of host memory");
  }
  auto bail = [&](int code, const std::string& msg) {
    g_multi_init_err = msg;
    rt_multi_shutdown(m);
    return code;
  };
  for (int i = 0; i < n_devices; i++) {
    Worker& w = m->w[(size_t)i];
    w.device = device_ids[i];
    int rc = rt_init(w.device, &w.ctx);
    if (rc != RT_OK) return bail(rc, std::string("rt_multi_init: ") + rt_last_error(nullptr));
    hipError_t e = hipSetDevice(w.device);
    if (e == hipSuccess) e = rti::stream_create(&w.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = rti::event_create(&w.done, hipEventDisableTiming);
    if (e != hipSuccess) return bail(RT_E_HIP, std::string("rt_multi_init: ") + hipGetErrorString(e));
  }
  // the root (first device) reads the other devices' partial frames in place: peer access root -> peer
  const int root = m->w[0].device;
  hipError_t e = hipSetDevice(root);
  if (e != hipSuccess) return bail(RT_E_HIP, std::string("rt_multi_init: ") + hipGetErrorString(e));
  for (int i = 1; i < n_devices; i++) {
    const int peer = m->w[(size_t)i].device;
    if (peer == root) continue;
    int can = 0;
    e = hipDeviceCanAccessPeer(&can, root, peer);
    if (e != hipSuccess || !can)
      return bail(RT_E_UNSUPPORTED, "rt_multi_init: device " + std::to_string(root) + " cannot access device " + std::to_string(peer) + " (no peer path)");
    e = hipDeviceEnablePeerAccess(peer, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
    if (e != hipSuccess) return bail(RT_E_HIP, std::string("rt_multi_init: hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
  }
  e = rti::event_create(&m->ev_t0);
  if (e == hipSuccess) e = rti::event_create(&m->ev_t1);
  if (e == hipSuccess) e = rti::event_create(&m->ev_t2);
  if (e != hipSuccess) return bail(RT_E_HIP, std::string("rt_multi_init: ") + hipGetErrorString(e));
  try {
    for (int i = 0; i < n_devices; i++) m->w[(size_t)i].thread = std::thread(worker_main, m, i);
  } catch (...) {
    return bail(RT_E_INTERNAL, "rt_multi_init: could not start the per-device host threads");
  }
  *out = m;
  return RT_OK;
}

int rt_multi_set_option(rt_multi* m, const char* key, int64_t value) {
  if (!m || !key) return RT_E_INVALID_ARG;
  for (Worker& w : m->w) {
    int rc = rt_set_option(w.ctx, key, value);
    if (rc != RT_OK) return mfail(m, rc, rt_last_error(w.ctx));
  }
  return RT_OK;
}

int rt_multi_scene_upload(rt_multi* m, const rt_scene_desc* desc) {
  if (!m) return RT_E_INVALID_ARG;
  if (!desc) return mfail(m, RT_E_INVALID_ARG, "rt_multi_scene_upload: NULL argument");
  m->kind = rt_multi::UPLOAD;
  m->desc = desc;
  int rc = dispatch(m);
  m->desc = nullptr;
  return rc;
}

// renders on every device and leaves the summed frame in m->d_total on the root device (root stream = w[0].stream)
static int render_sum(rt_multi* m, const rt_camera* cam, uint64_t seed, uint32_t sample_begin, uint32_t sample_end,
                      uint32_t sample_stride) {
  if (!cam) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render: NULL argument");
  if (sample_stride == 0) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render: sample_stride must be >= 1");
  if (cam->width == 0 || cam->height == 0) return mfail(m, RT_E_INVALID_ARG, "camera: zero image size");
  if ((uint64_t)cam->width * cam->height > 0x3FFFFFFFull) return mfail(m, RT_E_UNSUPPORTED, "camera: image too large");
  for (Worker& w : m->w)
    if (!w.scene) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render: no scene uploaded");
  const uint32_t n = (uint32_t)m->w.size();
  if ((uint64_t)sample_stride * n > 0xFFFFFFFFull) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render: sample_stride x devices overflows");
  const size_t n_val = (size_t)cam->width * cam->height * 3;
  Worker& root = m->w[0];
  hipError_t e = hipSetDevice(root.device);
  if (e == hipSuccess && m->total_cap < n_val) {
    (void)hipStreamSynchronize(root.stream);
    if (m->d_total) (void)rti::dev_free(m->d_total);
    m->d_total = nullptr; m->total_cap = 0;
    e = rti::dev_malloc(&m->d_total, n_val * sizeof(float));
    if (e == hipSuccess) m->total_cap = n_val;
  }
  if (e != hipSuccess) return mfail(m, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, std::string("rt_multi_render: ") + hipGetErrorString(e));
  // device i renders the sample indices sample_begin + (i + k * n) * sample_stride < sample_end
  m->kind = rt_multi::RENDER;
  Job jobs[RT_MULTI_MAX_DEVICES];
  for (uint32_t i = 0; i < n; i++) {
    const uint64_t b = (uint64_t)sample_begin + (uint64_t)i * sample_stride;
    jobs[i].cam = cam; jobs[i].seed = seed;
    jobs[i].begin = b > sample_end ? sample_end : (uint32_t)b;
    jobs[i].end = sample_end;
    jobs[i].stride = sample_stride * n;
  }
  m->job_table = jobs;
  const int rc = dispatch(m);
  m->job_table = nullptr;
  if (rc != RT_OK) return rc;
  // root stream: wait for every device's render, then one kernel sums the frames in list order
  e = hipSetDevice(root.device);
  for (uint32_t i = 1; i < n && e == hipSuccess; i++) e = hipStreamWaitEvent(root.stream, m->w[i].done, 0);
  if (e == hipSuccess) e = hipEventRecord(m->ev_t0, root.stream);
  if (e == hipSuccess) {
    SumParts parts;
    memset(&parts, 0, sizeof(parts));
    for (uint32_t i = 0; i < n; i++) parts.p[i] = m->w[i].d_part;
    const size_t n4 = n_val / 4 + 1;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n4 + 255) / 256, 4096);
    (void)hipGetLastError();  // the launch's own status, not a stale one of this thread
    hipLaunchKernelGGL(k_sum_frames, dim3(blocks), dim3(256), 0, root.stream, parts, (int)n, m->d_total, n_val);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(m->ev_t1, root.stream);
  if (e != hipSuccess) return mfail(m, RT_E_HIP, std::string("rt_multi_render: ") + hipGetErrorString(e));
  return RT_OK;
}

// after the root stream has been synchronised (it waited for every device's render): a frame that a scheduler watchdog
// cut short on any device is an error of the whole call
static int frames_complete(rt_multi* m) {
  for (Worker& w : m->w) {
    (void)hipSetDevice(w.device);
    const int rc = rt_internal_frame_status(w.ctx);
    if (rc != RT_OK) return mfail(m, rc, "device " + std::to_string(w.device) + ": " + rt_last_error(w.ctx));
  }
  (void)hipSetDevice(m->w[0].device);
  return RT_OK;
}

int rt_multi_render(rt_multi* m, const rt_camera* cam, uint64_t seed, uint32_t sample_begin, uint32_t sample_end,
                    uint32_t sample_stride, float* out_sum) {
  if (!m) return RT_E_INVALID_ARG;
  if (!out_sum) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render: NULL argument");
  int rc = render_sum(m, cam, seed, sample_begin, sample_end, sample_stride);
  if (rc != RT_OK) return rc;
  Worker& root = m->w[0];
  const size_t bytes = (size_t)cam->width * cam->height * 3 * sizeof(float);
  hipError_t e = hipMemcpyAsync(out_sum, m->d_total, bytes, hipMemcpyDeviceToHost, root.stream);
  if (e == hipSuccess) e = hipEventRecord(m->ev_t2, root.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
  if (e != hipSuccess) return mfail(m, RT_E_HIP, std::string("rt_multi_render: ") + hipGetErrorString(e));
  m->stats_valid = true;
  return frames_complete(m);
}

int rt_multi_render_device(rt_multi* m, const rt_camera* cam, uint64_t seed, uint32_t sample_begin, uint32_t sample_end,
                           uint32_t sample_stride, float* d_out_sum_root) {
  if (!m) return RT_E_INVALID_ARG;
  if (!d_out_sum_root) return mfail(m, RT_E_INVALID_ARG, "rt_multi_render_device: NULL argument");
  int rc = render_sum(m, cam, seed, sample_begin, sample_end, sample_stride);
  if (rc != RT_OK) return rc;
  Worker& root = m->w[0];
  const size_t bytes = (size_t)cam->width * cam->height * 3 * sizeof(float);
  hipError_t e = hipMemcpyAsync(d_out_sum_root, m->d_total, bytes, hipMemcpyDeviceToDevice, root.stream);
  if (e == hipSuccess) e = hipEventRecord(m->ev_t2, root.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
  if (e != hipSuccess) return mfail(m, RT_E_HIP, std::string("rt_multi_render_device: ") + hipGetErrorString(e));
  m->stats_valid = true;
  return frames_complete(m);
}

int rt_multi_get_stats(rt_multi* m, rt_multi_stats* out) {
  if (!m || !out) return RT_E_INVALID_ARG;
  memset(&m->stats, 0, sizeof(m->stats));
  m->stats.n_devices = (uint32_t)m->w.size();
  if (m->stats_valid) {
    (void)hipSetDevice(m->w[0].device);
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1) == hipSuccess) m->stats.ms_reduce = ms;
    if (hipEventElapsedTime(&ms, m->ev_t1, m->ev_t2) == hipSuccess) m->stats.ms_copy = ms;
    for (size_t i = 0; i < m->w.size() && i < RT_MULTI_MAX_DEVICES; i++) {
      rt_stats s;
      const int rc = rt_get_stats(m->w[i].ctx, &s);
      if (rc != RT_OK) return mfail(m, rc, "device " + std::to_string(m->w[i].device) + ": " + rt_last_error(m->w[i].ctx));
      m->stats.ms_render[i] = s.ms_total;
      m->stats.paths += s.paths;
      m->stats.rays += s.rays;
    }
  }
  *out = m->stats;
  return RT_OK;
}

}  // extern "C"
