// Multi-process test of the cross-GPU modular reduce choreography (sda_amd/csrc/comm_plan.hpp) on a box WITHOUT GPUs:
// the product binds the plan to RCCL point-to-point calls and the HIP modular-sum kernel (sda_comm.cpp); here the
// same plan runs between forked processes over socket pairs, with a checker reducer (unsigned __int128).  What is
// covered: exact slice sizes (ragged and empty slices), ordering, in-place gather, and the no-overflow property a
// u64 sum collective would violate.   usage: comm_plan_test <world> <len> <modulus>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <vector>

#include "../../sda_amd/csrc/comm_plan.hpp"

using namespace sda;

static uint64_t splitmix(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int64_t value(int rank, size_t i, uint64_t q, bool worst) {
    return worst ? (int64_t)(q - 1) : (int64_t)(splitmix(((uint64_t)rank << 40) ^ i) % q);
}

struct SocketTransport : Transport {
    int rank, world;
    std::vector<int> fd;                                  // fd[peer], -1 for self
    struct Op { bool send; const int64_t* sbuf; int64_t* rbuf; size_t count; int peer; };
    std::vector<Op> ops;
    std::vector<int64_t> self;                            // loop-back queue
    int group_start() override { ops.clear(); return 0; }
    int send(const int64_t* b, size_t n, int p) override { ops.push_back({true, b, nullptr, n, p}); return 0; }
    int recv(int64_t* b, size_t n, int p) override { ops.push_back({false, nullptr, b, n, p}); return 0; }
    static int xfer(int f, void* p, size_t bytes, bool wr) {
        char* c = static_cast<char*>(p);
        while (bytes) {
            ssize_t r = wr ? write(f, c, bytes) : read(f, c, bytes);
            if (r <= 0) return 1;
            c += r; bytes -= (size_t)r;
        }
        return 0;
    }
    int group_end() override {                            // messages are small: all sends fit the socket buffers
        for (const Op& o : ops)
            if (o.send) {
                if (o.peer == rank) self.insert(self.end(), o.sbuf, o.sbuf + o.count);
                else if (xfer(fd[o.peer], const_cast<int64_t*>(o.sbuf), o.count * 8, true)) return 100;
            }
        size_t self_pos = 0;
        for (const Op& o : ops)
            if (!o.send) {
                if (o.peer == rank) { for (size_t i = 0; i < o.count; ++i) o.rbuf[i] = self[self_pos++]; }
                else if (xfer(fd[o.peer], o.rbuf, o.count * 8, false)) return 101;
            }
        self.clear();
        return 0;
    }
};

struct CheckerReducer : Reducer {
    uint64_t q;
    int modsum(const int64_t* in, size_t parts, size_t stride, size_t len, int64_t* out) override {
        for (size_t i = 0; i < len; ++i) {
            unsigned __int128 acc = 0;
            for (size_t g = 0; g < parts; ++g) acc += (uint64_t)in[g * stride + i];
            out[i] = (int64_t)(uint64_t)(acc % q);
        }
        return 0;
    }
};

// error path: an operation that fails inside a group must not leave the group open (an ncclGroupStart without its
// ncclGroupEnd poisons the thread's next RCCL call) and the FIRST error is the one reported
struct FailingTransport : Transport {
    int fail_at, calls = 0, starts = 0, ends = 0, open = 0;
    bool end_fails = false;
    int group_start() override { ++starts; ++open; return 0; }
    int send(const int64_t*, size_t, int) override { return ++calls == fail_at ? 77 : 0; }
    int recv(int64_t*, size_t, int) override { return ++calls == fail_at ? 78 : 0; }
    int group_end() override { ++ends; --open; return end_fails && fail_at && calls >= fail_at ? 79 : 0; }   // fails after a failed operation
};
struct NullReducer : Reducer {
    int modsum(const int64_t*, size_t, size_t, size_t, int64_t*) override { return 0; }
};
static int error_path_checks() {
    const int world = 4;
    const size_t len = 10;
    std::vector<int64_t> partial(len, 1), recv(world * 3 + 1), mine(4), out(len);
    NullReducer red;
    for (int fail_at = 1; fail_at <= 16; ++fail_at) {         // 8 operations per group, two groups
        FailingTransport tr;
        tr.fail_at = fail_at;
        tr.end_fails = true;                                  // a failing group_end must not mask the earlier error
        const int st = modular_allreduce_plan(tr, red, 1, world, partial.data(), len, recv.data(), mine.data(), out.data());
        const int want = tr.calls % 2 ? 77 : 78;              // odd calls are sends, even calls are receives
        if (st != want || tr.open != 0 || tr.starts != tr.ends) {
            fprintf(stderr, "error path: failure at operation %d -> status %d (want %d), %d groups opened, %d closed\n", fail_at, st, want,
                    tr.starts, tr.ends);
            return 1;
        }
    }
    FailingTransport ok;
    ok.fail_at = 0;
    if (modular_allreduce_plan(ok, red, 1, world, partial.data(), len, recv.data(), mine.data(), out.data()) || ok.starts != 2 || ok.ends != 2) return 1;
    return 0;
}

static int run_rank(int rank, int world, size_t len, uint64_t q, bool worst, std::vector<int> fd) {
    std::vector<int64_t> partial(len ? len : 1), out(len ? len : 1, -1);
    for (size_t i = 0; i < len; ++i) partial[i] = value(rank, i, q, worst);
    const SlicePlan pl(world, len);
    std::vector<int64_t> recv((size_t)world * pl.seg + 1), mine(pl.seg + 1);
    SocketTransport tr;
    tr.rank = rank; tr.world = world; tr.fd = fd;
    CheckerReducer red;
    red.q = q;
    if (int st = modular_allreduce_plan(tr, red, rank, world, partial.data(), len, recv.data(), mine.data(), out.data())) return st;
    for (size_t i = 0; i < len; ++i) {
        unsigned __int128 acc = 0;
        for (int r = 0; r < world; ++r) acc += (uint64_t)value(r, i, q, worst);
        if ((uint64_t)out[i] != (uint64_t)(acc % q)) {
            fprintf(stderr, "rank %d: element %zu is %lld, expected %llu\n", rank, i, (long long)out[i], (unsigned long long)(uint64_t)(acc % q));
            return 2;
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s world len modulus\n", argv[0]); return 64; }
    const int world = atoi(argv[1]);
    const size_t len = strtoull(argv[2], nullptr, 10);
    const uint64_t q = strtoull(argv[3], nullptr, 10);
    // slice plan invariants
    const SlicePlan pl(world, len);
    size_t total = 0;
    for (int g = 0; g < world; ++g) { if (pl.offset(g) != total || pl.count(g) > pl.seg) return 3; total += pl.count(g); }
    if (total != len) return 3;
    if (error_path_checks()) return 5;
    for (int worst = 0; worst < 2; ++worst) {
        std::vector<std::vector<int>> fds(world, std::vector<int>(world, -1));
        for (int a = 0; a < world; ++a)
            for (int b = a + 1; b < world; ++b) {
                int sv[2];
                if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 4;
                fds[a][b] = sv[0]; fds[b][a] = sv[1];
            }
        std::vector<pid_t> kids;
        for (int r = 0; r < world; ++r) {
            pid_t p = fork();
            if (p == 0) {
                for (int a = 0; a < world; ++a)
                    for (int b = 0; b < world; ++b)
                        if (a != r && fds[a][b] >= 0) close(fds[a][b]);
                _exit(run_rank(r, world, len, q, worst != 0, fds[r]));
            }
            kids.push_back(p);
        }
        for (auto& row : fds) for (int f : row) if (f >= 0) close(f);
        int bad = 0;
        for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) bad = 1; }
        if (bad) { fprintf(stderr, "world %d len %zu worst %d: FAILED\n", world, len, worst); return 1; }
    }
    printf("comm plan ok: world %d, len %zu, modulus %llu\n", world, len, (unsigned long long)q);
    return 0;
}
