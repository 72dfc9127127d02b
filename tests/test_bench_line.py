"""bench.py's line, without a GPU: `flat_evidence` copies the scalars the claims rest on into `config` / `roofline` under short keys and keeps every
string of the objects the driver's record keeps under 100 characters; `step_algorithmic_bytes` / `forward_algorithmic_bytes`
are the byte counts DESIGN.md section 4 states."""
import bench


def test_algorithmic_bytes_match_design():
    E, N = 65280, 256
    per = bench.step_algorithmic_bytes(E, 4, 3)
    assert per == [(16 + 4 + 24) * E, (24 + 4 + 24 + 4) * E, (24 + 4 + 24 + 4) * E]            # 44 / 56 / 56 B per edge: the message launches
    assert bench.step_algorithmic_bytes(E, 4, 3, msg_only=False)[-1] == (24 + 4 + 4) * E       # the last step: 32 B per edge
    assert bench.step_algorithmic_bytes(E, 4, 3, e_bytes=12)[1] == (12 + 4 + 12 + 4) * E       # bf16 edge state
    total = bench.forward_algorithmic_bytes(N, E)
    assert total == (44 + 56 + 56 + 32 + 20) * E + N * 2048 * 4 + (1 << 20)


def test_flat_evidence_keys_and_string_lengths():
    long = "x" * 300
    res = {"config": {"workload": long, "mode": "graph_block", "ms_per_step_by_mode": {"eager": 0.03, "graph_block": 0.027}},
           "roofline": {"kernel": long, "frac": 0.09}, "cpu_baseline": {"sample": long, "value": 1.0},
           "config4_sharded": {"graphs_per_rank": 512, "ms_per_step": 0.48, "value": 1.7e10, "ms_per_step_eager": 0.48, "ms_per_step_graph": 0.49,
                               "graph_equals_eager_bitwise": True, "roofline_rank0": {"frac": 0.64, "avg_launch_us": 82.0}, "enc_frac_rank0": 0.5,
                               "forward_frac_rank0": 0.6, "kernels_us_rank0": {"enc_gemm": 135.0}},
           "config4_share": {"ms_per_step": 0.084, "union_ms_per_step": 0.48, "projected_8gpu_speedup": 5.7, "ms_per_step_eager": 0.084,
                             "ms_per_step_graph": 0.088, "projected_8gpu_speedup_eager": 5.7, "enc_frac": 0.32, "step_frac": 0.5, "forward_frac": 0.42,
                             "kernels_us": {"plan": 6.9, "enc_gemm": 26.0}},
           "roofline_at_scale": {"frac": 0.6, "avg_launch_us": 45.0},
           "terrace_pipeline": {"ms_per_batch": 0.12, "frames_per_s": 5e5, "final_over_chain": 1.4, "parity": {"ok": True},
                                "with_rounding_and_splitting": {"ms_per_batch": 0.6, "frames_per_s": 1e5, "overlapped_ms_per_batch": 0.17,
                                                                "overlapped_frames_per_s": 3.8e5, "frames_through_the_host_heuristics_per_batch": 58.4}},
           "train_step": {"ms_per_iteration": 0.43, "parity": {"grad_max_abs_err": 5e-8, "ok": True}},
           "configs": {"config2_dense64_L4_fp32": {"ms_per_step": 0.024, "parity": {"max_abs_err": [1e-8, 6e-8, 3e-8]}, "roofline": {"frac": 0.006}},
                       "config5_dense1024_L8_fp32": {"error": "RuntimeError: " + long}},
           "parity": {"max_abs_err": [3e-8, 3e-8, 3e-8], "ok": True},
           "dynamic_range": {"nodes": 64, "max_abs_logit": 79.4, "hip_fp32_state_rel": 8.7e-7, "ok": True}}
    out = bench.flat_evidence(res)
    c, r = out["config"], out["roofline"]
    for key, want in {"share_ms": 0.084, "union_ms": 0.48, "projected_8gpu_speedup": 5.7, "cfg4_ms": 0.48, "cfg4_ms_graph": 0.49, "terrace_ms_per_batch": 0.12,
                      "terrace_final_ms_per_batch": 0.6, "terrace_final_over_chain": 1.4, "terrace_final_overlapped_ms_per_batch": 0.17, "train_ms_per_iteration": 0.43, "cfg2_ms": 0.024,
                      "cfg2_err": 6e-8, "headline_err": 3e-8, "dynrange_max_abs_logit": 79.4, "ms_eager": 0.03, "ms_graph_block": 0.027}.items():
        assert c[key] == want, key
    for key, want in {"cfg4_step_frac": 0.64, "cfg4_enc_frac": 0.5, "cfg4_enc_us": 135.0, "share_enc_frac": 0.32, "share_plan_us": 6.9, "at_scale_frac": 0.6,
                      "cfg2_step_frac": 0.006}.items():
        assert r[key] == want, key
    assert "PROJECTION" in c["projected_8gpu_is"] and c["cfg5_error"].startswith("RuntimeError")
    for name in ("config", "roofline", "cpu_baseline"):
        for k, v in out[name].items():
            assert not isinstance(v, str) or len(v) <= 100, (name, k, len(v))
    assert out["notes"]["config.workload"] == long and out["notes"]["roofline.kernel"] == long      # the long forms are kept, in a nested object
