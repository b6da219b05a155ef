// samedec_main.cpp -- `samedec_gpu`: the reference's command-line decoder on top of the C ABI
// (SURVEY.md section 8f "next-2").
//
// Reads signed 16-bit native-endian PCM from --file or standard input, casts every sample to
// f32 unscaled, prints each decoded message on its own line ("ZCZC-..." or "NNNN") and, when a
// command follows "--", runs it for the duration of each message with the audio on its
// standard input and the SAMEDEC_* variables in its environment.  Same options, same state
// machine, same EOF flush as crates/samedec/src/{main.rs:29-49, cli.rs:48-139, app.rs:49-244,
// spawner.rs:24-77}; the receiver behind it is one channel of the MI355X library.
//
// Process layout.  The alert command is started by a small helper process that is forked
// before the GPU library is loaded (libsame_rx.so is dlopen'ed afterwards): a process that
// owns a GPU context never forks or execs.  The helper hands the write end of the child's
// stdin pipe back over a UNIX socket.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/same_place.h"
#include "../../include/same_rx.h"

namespace {

// ------------------------------------------------------------------------------------------
// command line (cli.rs:48-139)
// ------------------------------------------------------------------------------------------
struct Args {
    int verbose = 0;
    bool quiet = false;
    uint32_t rate = 22050;
    std::string file = "-";
    bool demo = false;
    float dc_blocker_len = 0.38f, agc_bw = 0.01f;
    float timing_bw_unlocked = 0.125f, timing_bw_locked = 0.05f, timing_max_dev = 0.01f;
    float squelch_pwr_open = 0.10f, squelch_pwr_close = 0.05f;
    uint32_t preamble_max_errors = 2;
    std::vector<std::string> child;
    std::string library;           // --library PATH (not in the reference): libsame_rx.so to load
    int device = 0;                // --device N (not in the reference)
};

int g_log_level = 1;   // 0 quiet, 1 warn, 2 info, 3 debug
void logmsg(int level, const char *tag, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
void logmsg(int level, const char *tag, const char *fmt, ...)
{
    if (level > g_log_level) return;
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, " %s samedec > ", tag);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}
#define LOG_ERROR(...) logmsg(1, "ERROR", __VA_ARGS__)
#define LOG_WARN(...) logmsg(1, "WARN ", __VA_ARGS__)
#define LOG_INFO(...) logmsg(2, "INFO ", __VA_ARGS__)
#define LOG_DEBUG(...) logmsg(3, "DEBUG", __VA_ARGS__)

const char *kUsage =
    "Usage: samedec_gpu [OPTIONS] [-- <CHILD>...]\n\n"
    "Arguments:\n  [CHILD]...  Child process and arguments, started for every message\n\n"
    "Options:\n"
    "  -v, --verbose...     Verbosity level (-vvv for more)\n"
    "  -q, --quiet          Disable all console output\n"
    "  -r, --rate <RATE>    Sampling rate (Hz) [default: 22050]\n"
    "      --file <FILE>    Input file (or \"-\" for stdin) [default: -]\n"
    "      --demo           Issue demo warning and exit\n"
    "  -h, --help           Print help\n"
    "  -V, --version        Print version\n\n"
    "Advanced Modem Options:\n"
    "      --dc-blocker-len <LEN>        DC-blocking filter length (fraction of baud rate) [default: 0.38]\n"
    "      --agc-bw <BW>                 AGC bandwidth (fraction of baud rate) [default: 0.01]\n"
    "      --timing-bw-unlocked <BW>     Timing loop bandwidth, before sync [default: 0.125]\n"
    "      --timing-bw-locked <BW>       Timing loop bandwidth, after sync [default: 0.05]\n"
    "      --timing-max-dev <DEV>        Maximum timing deviation (fraction of baud rate) [default: 0.01]\n"
    "      --squelch-pwr-open <PWR>      Squelch open power level [default: 0.1]\n"
    "      --squelch-pwr-close <PWR>     Squelch close power level [default: 0.05]\n"
    "      --preamble-max-errors <N>     Maximum preamble sync bit errors, 0-5 [default: 2]\n\n"
    "MI355X build:\n"
    "      --library <PATH>              libsame_rx.so to load [default: next to this program]\n"
    "      --device <N>                  GPU ordinal [default: 0]\n\n"
    "This program accepts raw PCM samples in signed 16-bit (i16) format, at the given sampling\n"
    "--rate, and decodes any SAME headers that are present. Decoded headers are printed in their\n"
    "ASCII representation.\n\nALWAYS TEST YOUR DECODING SETUP!\n";

[[noreturn]] void usage_error(const std::string &what)
{
    fprintf(stderr, "error: %s\n\n%s", what.c_str(), kUsage);
    exit(2);
}

bool parse_args(int argc, char **argv, Args *a)
{
    auto value = [&](int &i, const std::string &arg, const char *name, std::string *out) -> bool {
        const std::string eq = std::string(name) + "=";
        if (arg == name) {
            if (i + 1 >= argc) usage_error(std::string("a value is required for '") + name + "'");
            *out = argv[++i];
            return true;
        }
        if (arg.compare(0, eq.size(), eq) == 0) { *out = arg.substr(eq.size()); return true; }
        return false;
    };
    auto to_f = [&](const std::string &v, const char *name) {
        char *end = nullptr;
        const float f = strtof(v.c_str(), &end);
        if (v.empty() || *end) usage_error("invalid value '" + v + "' for '" + name + "'");
        return f;
    };
    auto to_u = [&](const std::string &v, const char *name) {
        char *end = nullptr;
        const unsigned long u = strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end || v[0] == '-' || u > 0xfffffffful) usage_error("invalid value '" + v + "' for '" + name + "'");
        return (uint32_t)u;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string arg = argv[i];
        std::string v;
        if (arg == "--") { for (int j = i + 1; j < argc; ++j) a->child.push_back(argv[j]); break; }
        else if (arg == "-h" || arg == "--help") { fputs(kUsage, stdout); exit(0); }
        else if (arg == "-V" || arg == "--version") { puts("samedec_gpu 0.6.0 (MI355X)"); exit(0); }
        else if (arg == "-q" || arg == "--quiet") a->quiet = true;
        else if (arg == "--verbose") a->verbose += 1;
        else if (arg.size() >= 2 && arg[0] == '-' && arg[1] == 'v' && arg.find_first_not_of('v', 1) == std::string::npos)
            a->verbose += (int)arg.size() - 1;
        else if (arg == "--demo") a->demo = true;
        else if (value(i, arg, "--rate", &v) || value(i, arg, "-r", &v)) a->rate = to_u(v, "--rate");
        else if (value(i, arg, "--file", &v)) a->file = v;
        else if (value(i, arg, "--dc-blocker-len", &v)) a->dc_blocker_len = to_f(v, "--dc-blocker-len");
        else if (value(i, arg, "--agc-bw", &v)) a->agc_bw = to_f(v, "--agc-bw");
        else if (value(i, arg, "--timing-bw-unlocked", &v)) a->timing_bw_unlocked = to_f(v, "--timing-bw-unlocked");
        else if (value(i, arg, "--timing-bw-locked", &v)) a->timing_bw_locked = to_f(v, "--timing-bw-locked");
        else if (value(i, arg, "--timing-max-dev", &v)) a->timing_max_dev = to_f(v, "--timing-max-dev");
        else if (value(i, arg, "--squelch-pwr-open", &v)) a->squelch_pwr_open = to_f(v, "--squelch-pwr-open");
        else if (value(i, arg, "--squelch-pwr-close", &v)) a->squelch_pwr_close = to_f(v, "--squelch-pwr-close");
        else if (value(i, arg, "--preamble-max-errors", &v)) {
            a->preamble_max_errors = to_u(v, "--preamble-max-errors");
            if (a->preamble_max_errors > 5) usage_error("invalid value '" + v + "' for '--preamble-max-errors': not in 0..6");
        }
        else if (value(i, arg, "--library", &v)) a->library = v;
        else if (value(i, arg, "--device", &v)) a->device = (int)to_u(v, "--device");
        else usage_error("unexpected argument '" + arg + "' found");
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// spawn helper: a process forked before the GPU library is loaded
// ------------------------------------------------------------------------------------------
// request  = 'S' u32 n_args (NUL-terminated strings) u32 n_env ("K=V" strings)   -> reply i32 pid|-errno (+ fd)
//            'W' i32 pid                                                           -> reply i32 wait status
bool write_all(int fd, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t w = write(fd, c, n);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        c += w; n -= (size_t)w;
    }
    return true;
}
bool read_all(int fd, void *p, size_t n)
{
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t r = read(fd, c, n);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;
        c += r; n -= (size_t)r;
    }
    return true;
}

int send_reply(int sock, int32_t value, int fd)
{
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    struct iovec iov = {&value, sizeof(value)};
    msg.msg_iov = &iov;
    msg.msg_iovlen = 1;
    char ctl[CMSG_SPACE(sizeof(int))];
    if (fd >= 0) {
        std::memset(ctl, 0, sizeof(ctl));
        msg.msg_control = ctl;
        msg.msg_controllen = sizeof(ctl);
        struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
        c->cmsg_level = SOL_SOCKET;
        c->cmsg_type = SCM_RIGHTS;
        c->cmsg_len = CMSG_LEN(sizeof(int));
        std::memcpy(CMSG_DATA(c), &fd, sizeof(int));
    }
    return sendmsg(sock, &msg, 0) == (ssize_t)sizeof(value) ? 0 : -1;
}

int recv_reply(int sock, int32_t *value, int *fd)
{
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    struct iovec iov = {value, sizeof(*value)};
    msg.msg_iov = &iov;
    msg.msg_iovlen = 1;
    char ctl[CMSG_SPACE(sizeof(int))];
    msg.msg_control = ctl;
    msg.msg_controllen = sizeof(ctl);
    ssize_t r;
    do { r = recvmsg(sock, &msg, MSG_CMSG_CLOEXEC); } while (r < 0 && errno == EINTR);
    if (r != (ssize_t)sizeof(*value)) return -1;
    *fd = -1;
    for (struct cmsghdr *c = CMSG_FIRSTHDR(&msg); c; c = CMSG_NXTHDR(&msg, c))
        if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS) std::memcpy(fd, CMSG_DATA(c), sizeof(int));
    return 0;
}

bool read_strings(int sock, std::vector<std::string> *out)
{
    uint32_t n = 0;
    if (!read_all(sock, &n, sizeof(n))) return false;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t len = 0;
        if (!read_all(sock, &len, sizeof(len))) return false;
        std::string s(len, '\0');
        if (len && !read_all(sock, &s[0], len)) return false;
        out->push_back(s);
    }
    return true;
}
bool write_strings(int sock, const std::vector<std::string> &v)
{
    const uint32_t n = (uint32_t)v.size();
    if (!write_all(sock, &n, sizeof(n))) return false;
    for (const std::string &s : v) {
        const uint32_t len = (uint32_t)s.size();
        if (!write_all(sock, &len, sizeof(len))) return false;
        if (len && !write_all(sock, s.data(), len)) return false;
    }
    return true;
}

[[noreturn]] void helper_main(int sock)
{
    signal(SIGPIPE, SIG_IGN);
    for (;;) {
        char op = 0;
        if (!read_all(sock, &op, 1)) _exit(0);            // parent is gone
        if (op == 'S') {
            std::vector<std::string> args, env;
            if (!read_strings(sock, &args) || !read_strings(sock, &env) || args.empty()) _exit(1);
            int in_pipe[2], err_pipe[2];
            if (pipe2(in_pipe, O_CLOEXEC) || pipe2(err_pipe, O_CLOEXEC)) { send_reply(sock, -errno, -1); continue; }
            const pid_t pid = fork();
            if (pid < 0) {
                send_reply(sock, -errno, -1);
                close(in_pipe[0]); close(in_pipe[1]); close(err_pipe[0]); close(err_pipe[1]);
                continue;
            }
            if (pid == 0) {
                // the alert command: stdin = the pipe, stdout/stderr inherited (spawner.rs:46-50)
                signal(SIGPIPE, SIG_DFL);
                dup2(in_pipe[0], 0);
                for (const std::string &kv : env) {
                    const size_t eq = kv.find('=');
                    setenv(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str(), 1);
                }
                std::vector<char *> argv;
                for (std::string &s : args) argv.push_back(&s[0]);
                argv.push_back(nullptr);
                execvp(argv[0], argv.data());
                const int e = errno;
                (void)!write(err_pipe[1], &e, sizeof(e));
                _exit(127);
            }
            close(in_pipe[0]);
            close(err_pipe[1]);
            int e = 0;
            const ssize_t r = read(err_pipe[0], &e, sizeof(e));   // EOF once exec succeeded
            close(err_pipe[0]);
            if (r == (ssize_t)sizeof(e)) {
                waitpid(pid, nullptr, 0);
                send_reply(sock, -e, -1);
            } else {
                send_reply(sock, (int32_t)pid, in_pipe[1]);
            }
            close(in_pipe[1]);
        } else if (op == 'W') {
            int32_t pid = 0;
            if (!read_all(sock, &pid, sizeof(pid))) _exit(1);
            int status = 0;
            pid_t r;
            do { r = waitpid(pid, &status, 0); } while (r < 0 && errno == EINTR);
            send_reply(sock, r < 0 ? -errno : (int32_t)status, -1);
        } else {
            _exit(1);
        }
    }
}

struct Child { int32_t pid = -1; int stdin_fd = -1; };

class Spawner {
public:
    bool start()
    {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, sv)) return false;
        const pid_t pid = fork();
        if (pid < 0) return false;
        if (pid == 0) { close(sv[0]); helper_main(sv[1]); }
        close(sv[1]);
        sock_ = sv[0];
        return true;
    }
    // spawner::spawn spawner.rs:24-77
    int spawn(const std::vector<std::string> &args, const std::vector<std::string> &env, Child *out)
    {
        const char op = 'S';
        if (!write_all(sock_, &op, 1) || !write_strings(sock_, args) || !write_strings(sock_, env)) return -EPIPE;
        int32_t v = 0; int fd = -1;
        if (recv_reply(sock_, &v, &fd)) return -EPIPE;
        if (v < 0) return v;
        out->pid = v; out->stdin_fd = fd;
        return 0;
    }
    int wait(int32_t pid, int *status)
    {
        const char op = 'W';
        if (!write_all(sock_, &op, 1) || !write_all(sock_, &pid, sizeof(pid))) return -EPIPE;
        int32_t v = 0; int fd = -1;
        if (recv_reply(sock_, &v, &fd)) return -EPIPE;
        if (v < 0) return v;
        *status = v;
        return 0;
    }
private:
    int sock_ = -1;
};

// ------------------------------------------------------------------------------------------
// the GPU library, loaded after the helper exists
// ------------------------------------------------------------------------------------------
struct Lib {
    decltype(&same_last_error) last_error;
    decltype(&same_rx_builder_new) builder_new;
    decltype(&same_rx_builder_free) builder_free;
    decltype(&same_rx_builder_with_agc_gain_limits) with_agc_gain_limits;
    decltype(&same_rx_builder_with_agc_bandwidth) with_agc_bandwidth;
    decltype(&same_rx_builder_with_dc_blocker_length) with_dc_blocker_length;
    decltype(&same_rx_builder_with_timing_bandwidth) with_timing_bandwidth;
    decltype(&same_rx_builder_with_timing_max_deviation) with_timing_max_deviation;
    decltype(&same_rx_builder_with_squelch_power) with_squelch_power;
    decltype(&same_rx_builder_with_preamble_max_errors) with_preamble_max_errors;
    decltype(&same_rx_build) rx_build;
    decltype(&same_rx_free) rx_free;
    decltype(&same_rx_process) rx_process;
    decltype(&same_rx_flush) rx_flush;
    decltype(&same_header_new_with_error_info) header_new;
    decltype(&same_header_originator_str) originator_str;
    decltype(&same_header_originator) originator;
    decltype(&same_originator_display_str) originator_display;
    decltype(&same_header_event_str) event_str;
    decltype(&same_header_event) event;
    decltype(&same_event_display) event_display;
    decltype(&same_significance_code_str) significance_code;
    decltype(&same_header_location_count) location_count;
    decltype(&same_header_location) location;
    decltype(&same_header_issue_datetime) issue_datetime;
    decltype(&same_header_purge_datetime) purge_datetime;
    decltype(&same_header_is_national) is_national;
};

template <typename F> bool sym(void *h, const char *name, F *out)
{
    *out = reinterpret_cast<F>(dlsym(h, name));
    if (!*out) fprintf(stderr, "samedec_gpu: %s is missing from the library\n", name);
    return *out != nullptr;
}

bool load_library(const Args &args, const char *argv0, Lib *L)
{
    std::string path = args.library;
    if (path.empty()) {
        char self[4096];
        const ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 1);
        std::string dir = n > 0 ? std::string(self, (size_t)n) : std::string(argv0);
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? "." : dir.substr(0, slash);
        path = dir + "/libsame_rx.so";
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "samedec_gpu: cannot load %s: %s\n", path.c_str(), dlerror()); return false; }
    bool ok = true;
#define SYM(field, name) ok &= sym(h, #name, &L->field)
    SYM(last_error, same_last_error);
    SYM(builder_new, same_rx_builder_new); SYM(builder_free, same_rx_builder_free);
    SYM(with_agc_gain_limits, same_rx_builder_with_agc_gain_limits);
    SYM(with_agc_bandwidth, same_rx_builder_with_agc_bandwidth);
    SYM(with_dc_blocker_length, same_rx_builder_with_dc_blocker_length);
    SYM(with_timing_bandwidth, same_rx_builder_with_timing_bandwidth);
    SYM(with_timing_max_deviation, same_rx_builder_with_timing_max_deviation);
    SYM(with_squelch_power, same_rx_builder_with_squelch_power);
    SYM(with_preamble_max_errors, same_rx_builder_with_preamble_max_errors);
    SYM(rx_build, same_rx_build); SYM(rx_free, same_rx_free);
    SYM(rx_process, same_rx_process); SYM(rx_flush, same_rx_flush);
    SYM(header_new, same_header_new_with_error_info);
    SYM(originator_str, same_header_originator_str); SYM(originator, same_header_originator);
    SYM(originator_display, same_originator_display_str);
    SYM(event_str, same_header_event_str); SYM(event, same_header_event);
    SYM(event_display, same_event_display); SYM(significance_code, same_significance_code_str);
    SYM(location_count, same_header_location_count); SYM(location, same_header_location);
    SYM(issue_datetime, same_header_issue_datetime); SYM(purge_datetime, same_header_purge_datetime);
    SYM(is_national, same_header_is_national);
#undef SYM
    return ok;
}

// ------------------------------------------------------------------------------------------
// input: i16 native-endian samples, buffered so that unconsumed samples can be re-presented
// ------------------------------------------------------------------------------------------
class Input {
public:
    explicit Input(FILE *f) : f_(f) {}
    // limit the total number of samples this source will still deliver (Iterator::take)
    void set_limit(uint64_t n) { limited_ = true; left_ = n; }
    void clear_limit() { limited_ = false; }
    // make at least one unconsumed sample available; false at end of input
    bool fill()
    {
        if (pos_ < f32_.size()) return true;
        f32_.clear(); i16_.clear(); pos_ = 0;
        if (eof_) return false;
        size_t want = kChunk;
        if (limited_) { if (left_ == 0) return false; if (left_ < want) want = (size_t)left_; }
        i16_.resize(want);
        const size_t got = fread(i16_.data(), sizeof(int16_t), want, f_);
        i16_.resize(got);
        if (got < want) eof_ = true;          // read_i16().ok()? ends the iterator, main.rs:47
        if (got == 0) return false;
        if (limited_) left_ -= got;
        f32_.resize(got);
        for (size_t i = 0; i < got; ++i) f32_[i] = (float)i16_[i];   // `sa as f32`, app.rs:112
        return true;
    }
    const float *data() const { return f32_.data() + pos_; }
    const int16_t *raw() const { return i16_.data() + pos_; }
    size_t size() const { return f32_.size() - pos_; }
    void consume(size_t n) { pos_ += n; }
    // a take() that ends early gives the unread part of its budget back to nobody: the
    // samples simply stay in the underlying source
private:
    static constexpr size_t kChunk = 16384;
    FILE *f_;
    std::vector<int16_t> i16_;
    std::vector<float> f32_;
    size_t pos_ = 0;
    bool eof_ = false, limited_ = false;
    uint64_t left_ = 0;
};

struct Message {
    int kind = 0;                 // SAME_MSG_START / SAME_MSG_END
    same_header hdr;
};

struct App {
    const Args &args;
    const Lib &L;
    same_rx *rx;
    Spawner &spawner;

    bool event_to_message(const same_rx_event &ev, Message *m) const
    {
        // SameReceiverEvent::into_message_ok: only Transport(Message(Ok(..))) passes
        if (ev.kind == SAME_TRANSPORT_MSG_END) { m->kind = SAME_MSG_END; return true; }
        if (ev.kind != SAME_TRANSPORT_MSG_START) return false;
        const size_t n = ev.len < sizeof(ev.bytes) ? ev.len : sizeof(ev.bytes);
        if (L.header_new(reinterpret_cast<const char *>(ev.bytes), n, nullptr, 0, nullptr, 0, &m->hdr) < 0) return false;
        m->hdr.voting_byte_count = ev.aux;
        m->hdr.parity_error_count = ev.aux2;
        m->kind = SAME_MSG_START;
        return true;
    }

    // receiver.iter_messages(input).next(): pull samples until a message comes out; every
    // consumed sample is also written to `tee_fd` when that is >= 0 (run_child's inspect())
    bool next_message(Input &in, int tee_fd, Message *m)
    {
        for (;;) {
            if (!in.fill()) return false;
            size_t used = 0;
            same_rx_event ev;
            const int got = L.rx_process(rx, in.data(), in.size(), &used, &ev);
            if (got < 0) { LOG_ERROR("receiver failed: %s", L.last_error()); exit(1); }
            if (tee_fd >= 0 && used) (void)write_all(tee_fd, in.raw(), used * sizeof(int16_t));   // errors suppressed
            in.consume(used);
            if (got == 1 && event_to_message(ev, m)) return true;
        }
    }

    bool flush(Message *m)
    {
        same_rx_event ev;
        const int got = L.rx_flush(rx, &ev);
        if (got < 0) { LOG_ERROR("receiver failed: %s", L.last_error()); exit(1); }
        return got == 1 && event_to_message(ev, m);
    }

    std::vector<std::string> child_env(const same_header &h) const
    {
        // spawner.rs:33-76, one variable per accessor
        char buf[512];
        std::vector<std::string> env;
        auto put = [&](const char *k, const std::string &v) { env.push_back(std::string(k) + "=" + v); };
        const int64_t now = (int64_t)time(nullptr);
        int64_t t = 0;
        std::string issue, purge;
        if (L.issue_datetime(&h, now, &t) == 0) issue = std::to_string(t);
        if (L.purge_datetime(&h, now, &t) == 0) purge = std::to_string(t);
        int ph = 0, sg = 0;
        L.event(&h, &ph, &sg);
        put("SAMEDEC_RATE", std::to_string(args.rate));
        put("SAMEDEC_MSG", std::string(h.text, h.len));
        L.originator_str(&h, buf); put("SAMEDEC_ORG", buf);
        put("SAMEDEC_ORIGINATOR", L.originator_display(L.originator(&h)));
        L.event_str(&h, buf); put("SAMEDEC_EVT", buf);
        const size_t n = L.event_display(ph, sg, 0, buf, sizeof(buf)); put("SAMEDEC_EVENT", std::string(buf, n));
        put("SAMEDEC_SIGNIFICANCE", L.significance_code(sg));
        put("SAMEDEC_SIG_NUM", std::to_string(sg));
        std::string locs;
        for (size_t i = 0, nl = L.location_count(&h); i < nl; ++i) {
            const size_t m = L.location(&h, i, buf, sizeof(buf));
            if (i) locs += ' ';
            locs.append(buf, m);
        }
        put("SAMEDEC_LOCATIONS", locs);
        put("SAMEDEC_ISSUETIME", issue);
        put("SAMEDEC_PURGETIME", purge);
        put("SAMEDEC_IS_NATIONAL", L.is_national(&h) ? "Y" : "");
        return env;
    }

    // State<Alerting>::until_message_end app.rs:129-193
    void until_message_end(Message first, Input &in)
    {
        bool have = true;
        Message msg = first;
        while (have) {
            have = false;
            if (!args.quiet) { printf("%s\n", msg.kind == SAME_MSG_START ? msg.hdr.text : "NNNN"); fflush(stdout); }
            if (msg.kind != SAME_MSG_START) break;                  // EndOfMessage -> Waiting
            if (args.child.empty()) { LOG_DEBUG("no child process to spawn"); return; }
            Child child;
            const int rc = spawner.spawn(args.child, child_env(msg.hdr), &child);
            if (rc < 0) { LOG_ERROR("unable to spawn child process: %s", strerror(-rc)); return; }
            LOG_DEBUG("spawned child process PID %d", child.pid);
            // run_child app.rs:200-232: stream audio to the child until the next message
            Message next;
            have = next_message(in, child.stdin_fd, &next);
            if (have && next.kind == SAME_MSG_START) LOG_WARN("received SAME start-of-message without end-of-message");
            close(child.stdin_fd);
            int status = 0;
            const int wrc = spawner.wait(child.pid, &status);
            if (wrc < 0) LOG_ERROR("unable to await child process exit: %s", strerror(-wrc));
            else if (WIFEXITED(status) && WEXITSTATUS(status) == 0) LOG_DEBUG("child process exited successfully");
            else LOG_WARN("child process exited abnormally with status %d", WIFEXITED(status) ? WEXITSTATUS(status) : 1);
            if (have) msg = next;
        }
    }

    // app::run app.rs:49-75
    void run(Input &in)
    {
        if (args.demo) {
            LOG_WARN("demonstration (--demo) mode: the following messages are NOT LIVE!");
            const time_t now = time(nullptr);
            struct tm tmv;
            gmtime_r(&now, &tmv);
            char text[96];
            snprintf(text, sizeof(text), "ZCZC-EAS-DMO-999000+0015-%03d%02d%02d-N0 CALL -", tmv.tm_yday + 1, tmv.tm_hour, tmv.tm_min);
            Message dmo;
            dmo.kind = SAME_MSG_START;
            if (L.header_new(text, strlen(text), nullptr, 0, nullptr, 0, &dmo.hdr) < 0) { LOG_ERROR("unable to create DMO message"); exit(1); }
            in.set_limit((uint64_t)args.rate * 8u);
            until_message_end(dmo, in);
            in.clear_limit();
            Message eom;
            eom.kind = SAME_MSG_END;
            for (int i = 0; i < 3; ++i) until_message_end(eom, in);
            return;
        }
        for (;;) {
            // State<Waiting>::until_message_start app.rs:99-117
            Message msg;
            if (!next_message(in, -1, &msg) && !flush(&msg)) return;
            until_message_end(msg, in);
        }
    }
};

}  // namespace

int main(int argc, char **argv)
{
    Args args;
    parse_args(argc, argv, &args);
    g_log_level = args.quiet ? 0 : 1 + (args.verbose > 2 ? 2 : args.verbose);
    signal(SIGPIPE, SIG_IGN);

    Spawner spawner;                       // before anything can touch the GPU
    if (!args.child.empty() && !spawner.start()) { perror("samedec_gpu: cannot start the spawn helper"); return 1; }

    FILE *f = stdin;
    if (args.file == "-") {
        LOG_INFO("SAME decoder reading standard input");
        if (isatty(0)) {
            fprintf(stderr, "error: cowardly refusing to read audio samples from a terminal.\n\n"
                            "Pipe a source of raw uncompressed audio from sox, parec, rtl_fm,\n"
                            "or similar into this program.\n");
            return 1;
        }
    } else {
        LOG_INFO("SAME decoder reading file: \"%s\"", args.file.c_str());
        f = fopen(args.file.c_str(), "rb");
        if (!f) { fprintf(stderr, "error: Unable to open --file \"%s\": %s\n", args.file.c_str(), strerror(errno)); return 1; }
    }

    Lib L;
    if (!load_library(args, argv[0], &L)) return 1;
    // main.rs:29-37: the decoder's receiver configuration
    same_rx_builder *b = L.builder_new(args.rate);
    L.with_agc_gain_limits(b, 1.0f / 32767.0f, 1.0f / 200.0f);
    L.with_agc_bandwidth(b, args.agc_bw);
    L.with_dc_blocker_length(b, args.dc_blocker_len);
    L.with_timing_bandwidth(b, args.timing_bw_unlocked, args.timing_bw_locked);
    L.with_timing_max_deviation(b, args.timing_max_dev);
    L.with_squelch_power(b, args.squelch_pwr_open, args.squelch_pwr_close);
    L.with_preamble_max_errors(b, args.preamble_max_errors);
    same_rx *rx = nullptr;
    const int rc = L.rx_build(b, args.device, &rx);
    L.builder_free(b);
    if (rc != SAME_OK) { fprintf(stderr, "samedec_gpu: cannot build the receiver (%d): %s\n", rc, L.last_error()); return 1; }

    Input in(f);
    App app{args, L, rx, spawner};
    app.run(in);
    L.rx_free(rx);
    if (f != stdin) fclose(f);
    return 0;
}
