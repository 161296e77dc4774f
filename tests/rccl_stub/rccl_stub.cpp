// TEST INFRASTRUCTURE: a stand-in for librccl.so that moves bytes between PROCESSES ON THE CPU through Unix-domain sockets, so that the
// product's compiled transport (kajiya_amd/csrc/split.cpp: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd / ncclAllGather on an
// ncclComm_t it bootstraps itself) can run end to end in tests/test_multigpu_emulated.py, where "device" memory is host memory (tests/hip_emu).
// Loaded through KJ_RCCL_LIB; never shipped, never linked. Only the entry points split.cpp resolves, with the semantics it relies on:
//   * point-to-point operations between a pair of ranks match in the order they were posted;
//   * operations posted inside a group start together at ncclGroupEnd, so a rank may post its sends before its receives to several peers
//     without deadlocking (here: every queued operation progresses through non-blocking sockets in one poll loop);
//   * the stream argument is ignored: the emulated HIP runtime executes stream work at the call, and so does this.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>
#include <fcntl.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

namespace {

struct Id128 { char b[128]; };

struct Op { bool send; char* p; size_t left; char* base; };

struct Comm {
    int world = 0, rank = 0, listen_fd = -1;
    std::vector<int> fd;                       // per peer
    std::vector<std::deque<Op>> sends, recvs;  // per peer, posting order
    std::string path;
};

thread_local int g_depth = 0;
thread_local std::vector<Comm*> g_touched;

size_t type_size(int t) { return t <= 1 ? 1 : t <= 3 ? 4 : t <= 5 ? 8 : t == 6 ? 2 : t == 7 ? 4 : t == 8 ? 8 : 2; }   // ncclDataType_t

std::string sock_path(const Id128& id, int rank) {
    char buf[108];
    snprintf(buf, sizeof buf, "/tmp/kj_rccl_stub_%.32s_%d", id.b, rank);
    return buf;
}

bool write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) { ssize_t k = write(fd, c, n); if (k < 0) { if (errno == EINTR) continue; return false; } c += k; n -= size_t(k); }
    return true;
}
bool read_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) { ssize_t k = read(fd, c, n); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } c += k; n -= size_t(k); }
    return true;
}

// run every queued operation of `c` to completion
int progress(Comm* c) {
    for (;;) {
        std::vector<pollfd> pf;
        std::vector<int> peer;
        for (int p = 0; p < c->world; ++p) {
            short ev = 0;
            if (!c->sends[p].empty()) ev |= POLLOUT;
            if (!c->recvs[p].empty()) ev |= POLLIN;
            if (ev) { pf.push_back({c->fd[p], ev, 0}); peer.push_back(p); }
        }
        if (pf.empty()) return 0;
        static const int patience_ms = getenv("KJ_RCCL_STUB_TIMEOUT_MS") ? atoi(getenv("KJ_RCCL_STUB_TIMEOUT_MS")) : 60000;     // (real RCCL would wait forever)
        if (poll(pf.data(), pf.size(), patience_ms) <= 0) { fprintf(stderr, "[rccl stub] rank %d: no progress for %d ms\n", c->rank, patience_ms); return 1; }
        for (size_t i = 0; i < pf.size(); ++i) {
            const int p = peer[i];
            if ((pf[i].revents & POLLOUT) && !c->sends[p].empty()) {
                Op& o = c->sends[p].front();
                ssize_t k = o.left ? send(c->fd[p], o.p, o.left, MSG_DONTWAIT | MSG_NOSIGNAL) : 0;
                if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return 2;
                if (k > 0) { o.p += k; o.left -= size_t(k); }
                if (!o.left) c->sends[p].pop_front();
            }
            if ((pf[i].revents & (POLLIN | POLLHUP)) && !c->recvs[p].empty()) {
                Op& o = c->recvs[p].front();
                ssize_t k = o.left ? recv(c->fd[p], o.p, o.left, MSG_DONTWAIT) : 0;
                if (k == 0 && o.left) return 3;       // the peer went away
                if (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) return 2;
                if (k > 0) { o.p += k; o.left -= size_t(k); }
                if (!o.left) {
                    // KJ_RCCL_STUB_CORRUPT=<rank>: that rank's received messages arrive damaged (the self-test of the transport must notice)
                    const char* bad = getenv("KJ_RCCL_STUB_CORRUPT");
                    if (bad && atoi(bad) == c->rank && o.p != o.base) o.base[(o.p - o.base) / 2] ^= 0x5a;
                    c->recvs[p].pop_front();
                }
            }
        }
    }
}

int post(Comm* c, bool is_send, void* buf, size_t bytes, int peer) {
    if (!c || peer < 0 || peer >= c->world || peer == c->rank) return 4;
    (is_send ? c->sends : c->recvs)[peer].push_back(Op{is_send, (char*)buf, bytes, (char*)buf});
    if (g_depth == 0) return progress(c);
    bool seen = false;
    for (Comm* t : g_touched) seen |= t == c;
    if (!seen) g_touched.push_back(c);
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* out) {
    Id128* id = (Id128*)out;
    memset(id, 0, sizeof *id);
    timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->b, 33, "%08x%08x%08x", unsigned(getpid()), unsigned(ts.tv_sec), unsigned(ts.tv_nsec));
    return 0;
}

int ncclCommInitRank(void** out, int nranks, Id128 id, int rank) {
    Comm* c = new Comm();
    c->world = nranks; c->rank = rank;
    c->fd.assign(nranks, -1); c->sends.resize(nranks); c->recvs.resize(nranks);
    c->path = sock_path(id, rank);
    unlink(c->path.c_str());
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a{}; a.sun_family = AF_UNIX; strncpy(a.sun_path, c->path.c_str(), sizeof a.sun_path - 1);
    if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr*)&a, sizeof a) != 0 || listen(c->listen_fd, nranks) != 0) return 5;
    // connect to every lower rank (it may not be listening yet: retry), accept from every higher one; the connecting side names itself
    for (int p = 0; p < rank; ++p) {
        const std::string pp = sock_path(id, p);
        sockaddr_un b{}; b.sun_family = AF_UNIX; strncpy(b.sun_path, pp.c_str(), sizeof b.sun_path - 1);
        int fd = -1;
        for (int tries = 0; tries < 3000; ++tries) {
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (connect(fd, (sockaddr*)&b, sizeof b) == 0) break;
            close(fd); fd = -1;
            usleep(10000);
        }
        if (fd < 0) return 6;
        int32_t me = rank;
        if (!write_all(fd, &me, 4)) return 6;
        c->fd[p] = fd;
    }
    for (int k = rank + 1; k < nranks; ++k) {
        int fd = accept(c->listen_fd, nullptr, nullptr);
        int32_t who = -1;
        if (fd < 0 || !read_all(fd, &who, 4) || who <= rank || who >= nranks || c->fd[who] != -1) return 7;
        c->fd[who] = fd;
    }
    *out = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    for (int fd : c->fd) if (fd >= 0) close(fd);
    if (c->listen_fd >= 0) close(c->listen_fd);
    unlink(c->path.c_str());
    delete c;
    return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() {
    if (g_depth <= 0) return 8;
    if (--g_depth) return 0;
    int rc = 0;
    for (Comm* c : g_touched) { const int r = progress(c); if (r) rc = r; }
    g_touched.clear();
    return rc;
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void*) { return post((Comm*)comm, true, (void*)buf, count * type_size(dtype), peer); }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void*) { return post((Comm*)comm, false, buf, count * type_size(dtype), peer); }

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void*) {
    Comm* c = (Comm*)comm;
    if (!c) return 4;
    const size_t n = count * type_size(dtype);
    memmove((char*)recv + size_t(c->rank) * n, send, n);
    ncclGroupStart();
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank) {
            post(c, true, (char*)recv + size_t(c->rank) * n, n, p);
            post(c, false, (char*)recv + size_t(p) * n, n, p);
        }
    return ncclGroupEnd();
}

const char* ncclGetErrorString(int e) { return e == 0 ? "ok" : "rccl stub: transport failure"; }

}  // extern "C"
