"""Calls every export of libsimpleicp_hip.so with NULL context / NULL pointers and prints `name rc message`; run as a
child process by tests/test_host_api.py (a crash there is a failed test, not a dead test run)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from simpleicp_amd import _lib
L=_lib.load()
null=None
P=_lib.IterParams(); R=_lib.IterResult()
i64=C.c_int64(); d=C.c_double(); ci=C.c_int(); vp=C.c_void_p()
buf=(C.c_double*64)()
calls = {
 "sicp_ctx_destroy": lambda: L.sicp_ctx_destroy(null),
 "sicp_ctx_device_name": lambda: L.sicp_ctx_device_name(null, C.create_string_buffer(8), 8),
 "sicp_cloud_upload": lambda: L.sicp_cloud_upload(null, 0, buf, 1, 0),
 "sicp_cloud_upload_columns": lambda: L.sicp_cloud_upload_columns(null, 0, buf, buf, buf, 1, 0),
 "sicp_cloud_size": lambda: L.sicp_cloud_size(null, 0, C.byref(i64)),
 "sicp_cloud_transform": lambda: L.sicp_cloud_transform(null, 0, buf),
 "sicp_cloud_download": lambda: L.sicp_cloud_download(null, 0, buf),
 "sicp_cloud_download_columns": lambda: L.sicp_cloud_download_columns(null, 0, buf, buf, buf),
 "sicp_cloud_set_planarity": lambda: L.sicp_cloud_set_planarity(null, 0, null, null, 0, 0),
 "sicp_knn": lambda: L.sicp_knn(null, 0, buf, 1, 1, null, 1.0, buf, buf),
 "sicp_select_in_range": lambda: L.sicp_select_in_range(null, 0, 1, null, 0, null, 1.0, buf),
 "sicp_estimate_normals": lambda: L.sicp_estimate_normals(null, 0, buf, 1, 3, buf, buf, null),
 "sicp_icp_setup": lambda: L.sicp_icp_setup(null, buf, 1, buf, buf),
 "sicp_icp_iterate": lambda: L.sicp_icp_iterate(null, C.byref(P), C.byref(R)),
 "sicp_icp_run": lambda: L.sicp_icp_run(null, C.byref(P), 3, 1.0, C.byref(R), C.byref(i64)),
 "sicp_icp_get_state": lambda: L.sicp_icp_get_state(null, null, null, null, null),
 "sicp_icp_uncertainties": lambda: L.sicp_icp_uncertainties(null, buf),
 "sicp_icp_normal_equations": lambda: L.sicp_icp_normal_equations(null, buf, buf),
 "sicp_corr_match": lambda: L.sicp_corr_match(null, null, null, null),
 "sicp_corr_reject_planarity": lambda: L.sicp_corr_reject_planarity(null, 0.3, null, null, C.byref(i64)),
 "sicp_corr_reject_distances": lambda: L.sicp_corr_reject_distances(null, C.byref(d), C.byref(d), C.byref(i64)),
 "sicp_estimate_parameters": lambda: L.sicp_estimate_parameters(null, C.byref(P), null, C.byref(R)),
 "sicp_params_to_H": lambda: L.sicp_params_to_H(null, null),
 "sicp_set_exchange": lambda: L.sicp_set_exchange(null, _lib.EXCHANGE_FN(0), null, 0, 1, 0),
 "sicp_comm_unique_id": lambda: L.sicp_comm_unique_id(null),
 "sicp_comm_init": lambda: L.sicp_comm_init(null, null, 0, 1, 0),
 "sicp_comm_destroy": lambda: L.sicp_comm_destroy(null),
 "sicp_set_partition": lambda: L.sicp_set_partition(null, 0),
 "sicp_ctx_stream": lambda: L.sicp_ctx_stream(null, C.byref(vp)),
 "sicp_lexmin_gathered": lambda: L.sicp_lexmin_gathered(null, null, 1, 1, null, null, null),
 "sicp_timing_enable": lambda: L.sicp_timing_enable(null, 1),
 "sicp_timing_reset": lambda: L.sicp_timing_reset(null),
 "sicp_timing_get": lambda: L.sicp_timing_get(null, 0, C.byref(d), C.byref(i64)),
 "sicp_match_work": lambda: L.sicp_match_work(null, buf),
 "sicp_last_match_kernel": lambda: L.sicp_last_match_kernel(null, C.byref(ci)),
 "sicp_xyz_count": lambda: L.sicp_xyz_count(null, C.byref(i64)),
 "sicp_xyz_read": lambda: L.sicp_xyz_read(null, buf, 1, C.byref(i64), 1),
 "sicp_xyz_write": lambda: L.sicp_xyz_write(null, buf, 1, 3, 3, null, 1),
 "sicp_device_count": lambda: L.sicp_device_count(null),
 "sicp_ctx_create": lambda: L.sicp_ctx_create(0, null),
}
for name in (sys.argv[1:] or calls):
    rc = calls[name]()
    print(name, rc, L.sicp_last_error().decode()[:60], flush=True)
