// serve.h — `inquistr serve`: one process that keeps the device context, and the few lines `inquistr call` needs to hand
// its work to it (INQ_SERVER=<socket>).  Not part of the reference's CLI and not part of the C ABI: a deployment mode of the
// same `call`, for pipelines that start one process per sample (DESIGN.md §4 "a resident context").
#pragma once
#include <string>

#include "../../include/inquistr_host.h"

namespace inq {

// Runs until SIGTERM / SIGINT, a quit request, or `idle_exit_s` seconds (> 0) without a request.  Returns the exit status.
int serve_main(const char *socket_path, int device, double idle_exit_s);

// Hands one `call` to the server behind `socket_path`: the rows are written to `out_fd` BY THE SERVER (the descriptor travels
// with the request).  Returns 1 when the server answered (status / message filled in), 0 when there is no server to talk to
// (nothing was sent: the caller runs the call itself), -1 when the server went away after it had the request (rows may
// have been written: an error, not a case for doing the call again).
int client_call(const char *socket_path, const inq_call_args_t *a, int out_fd, int *status, std::string *message);

// INQ_SERVER=auto: the socket of this user's server for `device` (under $XDG_RUNTIME_DIR or /tmp), and a server started behind it
// when none answers - detached, leaving by itself after `idle_exit_s` seconds without a call.  True when a server listens (was
// there or came up within a few seconds).
std::string auto_socket_path(int device);
bool ensure_server(const char *self_exe, const char *socket_path, int device, double idle_exit_s);

// Asks the server to leave.  True if it answered.
bool client_quit(const char *socket_path);

}  // namespace inq
