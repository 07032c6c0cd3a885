"""Calls every export of libsimpleicp_hip.so with a NULL context / NULL pointers.  Run as a child process by
tests/test_host_api.py (a crash there is a failed test, not a dead test run) -- prints `name rc message` per export --
and imported by tests/native/asan_exercise.py, which runs the same calls under the sanitizers."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from simpleicp_amd import _lib  # noqa: E402


def build_calls():
    L = _lib.load()
    P, R = _lib.IterParams(), _lib.IterResult()
    i64, d, ci, vp = C.c_int64(), C.c_double(), C.c_int(), C.c_void_p()
    buf = (C.c_double * 64)()
    calls = {
        "sicp_device_count": lambda: L.sicp_device_count(None),
        "sicp_ctx_create": lambda: L.sicp_ctx_create(0, None),
        "sicp_ctx_destroy": lambda: L.sicp_ctx_destroy(None),
        "sicp_ctx_device_name": lambda: L.sicp_ctx_device_name(None, C.create_string_buffer(8), 8),
        "sicp_cloud_upload": lambda: L.sicp_cloud_upload(None, 0, buf, 1, 0),
        "sicp_cloud_upload_columns": lambda: L.sicp_cloud_upload_columns(None, 0, buf, buf, buf, 1, 0),
        "sicp_cloud_upload_start": lambda: L.sicp_cloud_upload_start(None, 0, buf, None, None, None, 1, 0),
        "sicp_cloud_upload_wait": lambda: L.sicp_cloud_upload_wait(None, 0),
        "sicp_cloud_size": lambda: L.sicp_cloud_size(None, 0, C.byref(i64)),
        "sicp_cloud_transform": lambda: L.sicp_cloud_transform(None, 0, buf),
        "sicp_cloud_download": lambda: L.sicp_cloud_download(None, 0, buf),
        "sicp_cloud_download_columns": lambda: L.sicp_cloud_download_columns(None, 0, buf, buf, buf),
        "sicp_cloud_download_both": lambda: L.sicp_cloud_download_both(None, 0, buf, buf, buf, buf),
        "sicp_cloud_set_planarity": lambda: L.sicp_cloud_set_planarity(None, 0, None, None, 0, 0),
        "sicp_knn": lambda: L.sicp_knn(None, 0, buf, 1, 1, None, 1.0, buf, buf),
        "sicp_select_in_range": lambda: L.sicp_select_in_range(None, 0, 1, None, 0, None, 1.0, buf),
        "sicp_estimate_normals": lambda: L.sicp_estimate_normals(None, 0, buf, 1, 3, buf, buf, None),
        "sicp_icp_setup": lambda: L.sicp_icp_setup(None, buf, 1, buf, buf),
        "sicp_icp_iterate": lambda: L.sicp_icp_iterate(None, C.byref(P), C.byref(R)),
        "sicp_icp_run": lambda: L.sicp_icp_run(None, C.byref(P), 3, 1.0, C.byref(R), C.byref(i64)),
        "sicp_icp_get_state": lambda: L.sicp_icp_get_state(None, None, None, None, None),
        "sicp_icp_uncertainties": lambda: L.sicp_icp_uncertainties(None, buf),
        "sicp_icp_normal_equations": lambda: L.sicp_icp_normal_equations(None, buf, buf),
        "sicp_corr_match": lambda: L.sicp_corr_match(None, None, None, None),
        "sicp_corr_reject_planarity": lambda: L.sicp_corr_reject_planarity(None, 0.3, None, None, C.byref(i64)),
        "sicp_corr_reject_distances": lambda: L.sicp_corr_reject_distances(None, C.byref(d), C.byref(d), C.byref(i64)),
        "sicp_estimate_parameters": lambda: L.sicp_estimate_parameters(None, C.byref(P), None, C.byref(R)),
        "sicp_params_to_H": lambda: L.sicp_params_to_H(None, None),
        "sicp_set_exchange": lambda: L.sicp_set_exchange(None, _lib.EXCHANGE_FN(0), None, 0, 1, 0),
        "sicp_comm_unique_id": lambda: L.sicp_comm_unique_id(None),
        "sicp_comm_init": lambda: L.sicp_comm_init(None, None, 0, 1, 0),
        "sicp_comm_destroy": lambda: L.sicp_comm_destroy(None),
        "sicp_comm_activate": lambda: L.sicp_comm_activate(None, 1, 0),
        "sicp_comm_info": lambda: L.sicp_comm_info(None, None),
        "sicp_device_memory": lambda: L.sicp_device_memory(None, None, None),
        "sicp_set_partition": lambda: L.sicp_set_partition(None, 0),
        "sicp_ctx_stream": lambda: L.sicp_ctx_stream(None, C.byref(vp)),
        "sicp_lexmin_gathered": lambda: L.sicp_lexmin_gathered(None, None, 1, 1, None, None, None),
        "sicp_timing_enable": lambda: L.sicp_timing_enable(None, 1),
        "sicp_timing_reset": lambda: L.sicp_timing_reset(None),
        "sicp_timing_get": lambda: L.sicp_timing_get(None, 0, C.byref(d), C.byref(i64)),
        "sicp_match_work": lambda: L.sicp_match_work(None, buf),
        "sicp_match_deferred": lambda: L.sicp_match_deferred(None, C.cast(buf, C.POINTER(C.c_uint64))),
        "sicp_tail_cycles": lambda: L.sicp_tail_cycles(None, buf),
        "sicp_tail_selection": lambda: L.sicp_tail_selection(None, buf),
        "sicp_exchange_info": lambda: L.sicp_exchange_info(None, buf),
        "sicp_knn_work": lambda: L.sicp_knn_work(None, buf),
        "sicp_last_match_kernel": lambda: L.sicp_last_match_kernel(None, C.byref(ci)),
        "sicp_xyz_count": lambda: L.sicp_xyz_count(None, C.byref(i64)),
        "sicp_xyz_read": lambda: L.sicp_xyz_read(None, buf, 1, C.byref(i64), 1),
        "sicp_xyz_write": lambda: L.sicp_xyz_write(None, buf, 1, 3, 3, None, 1),
    }
    return L, calls


def probe(names=None):
    """name -> (return code, message) of every export called with NULL arguments"""
    L, calls = build_calls()
    res = {}
    for name in (names or calls):
        rc = calls[name]()
        res[name] = (rc, L.sicp_last_error().decode())
    return res


if __name__ == "__main__":
    for name, (rc, msg) in probe(sys.argv[1:]).items():
        print(name, rc, msg[:60], flush=True)
