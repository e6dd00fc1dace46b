// Part of capi.cpp (ONE translation unit: this file is #included there and is not compiled on its own): the workspace pool of a handle.

// ---- workspace pool ----
Workspace* acquire_host(tkamd_tokenizer* t) {
    std::unique_lock<std::mutex> lk(t->mu);
    for (;;) {
        size_t n_host = 0;
        for (auto& w : t->pool) {
            if (w->device_bound) continue;
            ++n_host;
            if (!w->busy) { w->busy = true; t->last_used = w.get(); return w.get(); }
        }
        if (n_host < MAX_HOST_WORKSPACES) {
            t->pool.emplace_back(new Workspace());
            Workspace* w = t->pool.back().get();
            w->busy = true;
            t->last_used = w;
            return w;
        }
        t->cv.wait(lk);
    }
}
void release_host(tkamd_tokenizer* t, Workspace* w) {
    { std::lock_guard<std::mutex> lk(t->mu); w->busy = false; }
    t->cv.notify_one();
}
struct HostLease {
    tkamd_tokenizer* t;
    Workspace* w;
    HostLease(tkamd_tokenizer* t_) : t(t_), w(acquire_host(t_)) {}
    ~HostLease() { release_host(t, w); }
    HostLease(const HostLease&) = delete;
    HostLease& operator=(const HostLease&) = delete;
};
hipStream_t own_stream(Workspace* w) {
    if (!w->own_stream) HIP_CHECK(hipStreamCreateWithFlags(&w->own_stream, hipStreamNonBlocking));
    return w->own_stream;
}
// the device entry keeps one workspace per caller stream: the results of a call stay valid in it until the next call on that stream
Workspace* workspace_of_stream(tkamd_tokenizer* t, hipStream_t st, bool create) {
    std::lock_guard<std::mutex> lk(t->mu);
    for (auto& w : t->pool)
        if (w->device_bound && w->bound_stream == st) { t->last_used = w.get(); return w.get(); }
    if (!create) return nullptr;
    t->pool.emplace_back(new Workspace());
    Workspace* w = t->pool.back().get();
    w->device_bound = true;
    w->bound_stream = st;
    t->last_used = w;
    return w;
}

}  // namespace
