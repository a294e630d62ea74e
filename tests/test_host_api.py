"""Host logic of the drop-in model object that needs no GPU: validation rules and error texts of
the reference boundary (JIT!/vad/model/vad_annotator.py:17,91-127; utils_vad.py:33-49)."""
import pytest
import torch


class _NoEngine:
    device = 0


@pytest.fixture()
def model(built):
    from silero_vad_amd.engine import HipSileroVAD
    m = HipSileroVAD.__new__(HipSileroVAD)
    m.engine = _NoEngine()
    m.device = torch.device("cpu")
    m.sample_rates = [8000, 16000]
    m.reset_states()
    return m


def test_validate_input_rules(model):
    x, sr = model._validate_input(torch.zeros(512), 16000)
    assert x.shape == (1, 512) and sr == 16000
    x, sr = model._validate_input(torch.zeros(3, 1536), 48000)          # ::3 decimation
    assert x.shape == (3, 512) and sr == 16000
    x, sr = model._validate_input(torch.zeros(256), 8000)
    assert x.shape == (1, 256) and sr == 8000
    with pytest.raises(ValueError, match=r"Too many dimensions for input audio chunk 3"):
        model._validate_input(torch.zeros(1, 1, 512), 16000)
    with pytest.raises(ValueError, match=r"Supported sampling rates: \[8000, 16000\] \(or multiply of 16000\)"):
        model._validate_input(torch.zeros(512), 22050)
    with pytest.raises(ValueError, match="Input audio chunk is too short"):
        model._validate_input(torch.zeros(400), 16000)                   # 16000/400 = 40 > 31.25
    model._validate_input(torch.zeros(512), 16000)                       # 31.25 exactly is allowed


def test_call_rejects_wrong_chunk_size(model):
    with pytest.raises(ValueError, match=r"Provided number of samples is 640 \(Supported values: 256 for 8000 "
                                         r"sample rate, 512 for 16000\)"):
        model(torch.zeros(640), 16000)
    with pytest.raises(ValueError, match="Provided number of samples is 512"):
        model(torch.zeros(512), 8000)


def test_reset_states_protocol(model):
    model._last_sr, model._last_batch_size = 16000, 4
    model.reset_states()
    assert model._last_sr == 0 and model._last_batch_size == 0
    assert len(model._state) == 0 and len(model._context) == 0


def test_get_speech_timestamps_argument_errors(built):
    from silero_vad_amd import VADIterator, get_speech_timestamps

    class Dummy:
        def reset_states(self):
            pass

    with pytest.raises(ValueError, match="More than one dimension"):
        get_speech_timestamps(torch.zeros(2, 1024), Dummy())
    with pytest.raises(ValueError, match="Currently silero VAD models support 8000 and 16000"):
        get_speech_timestamps(torch.zeros(1024), Dummy(), sampling_rate=44100)
    with pytest.raises(ValueError, match="does not support sampling rates other than"):
        VADIterator(Dummy(), sampling_rate=48000)
    assert get_speech_timestamps(torch.zeros(0), Dummy()) == []


def test_hubconf_contract(built, tmp_path):
    """torch.hub entry (reference hubconf.py:26-56): `silero_vad()` -> (model, utils) with the reference's five utils in its order;
    without a GPU the model refuses loudly (no CPU fallback), the utils and the argument check work."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("hubconf", Path(__file__).resolve().parents[1] / "hubconf.py")
    hub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hub)
    assert "torch" in hub.dependencies
    with pytest.raises(Exception, match="Available ONNX opset_version"):
        hub.silero_vad(onnx=True, opset_version=14)
    if torch.cuda.is_available():
        model, utils = hub.silero_vad()
        assert model.sample_rates == [8000, 16000]
    else:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            hub.silero_vad()
        from silero_vad_amd import VADIterator, collect_chunks, get_speech_timestamps, read_audio, save_audio
        utils = (get_speech_timestamps, save_audio, read_audio, VADIterator, collect_chunks)
    assert [u.__name__ for u in utils] == ["get_speech_timestamps", "save_audio", "read_audio", "VADIterator", "collect_chunks"]
    x = 0.5 * torch.sin(torch.arange(8000) / 15.0)
    utils[1](str(tmp_path / "a.wav"), x, 8000)
    y = utils[2](str(tmp_path / "a.wav"), 8000)
    assert y.shape == x.shape and (x - y).abs().max() < 1.0 / 16384
    assert torch.equal(utils[4]([{"start": 10, "end": 20}, {"start": 100, "end": 130}], x), torch.cat([x[10:20], x[100:130]]))


def _worst_case_bench_record(world=8):
    """A full bench record as bench.main() holds it before printing, at its LARGEST: nine legs with their workload prose, parity
    records, roofline with every detail key, the corpus leg's five routes, cpu_baseline with five protocols, and the per-rank records of
    an 8-rank run.  Synthetic numbers; the shapes are the ones bench.py's legs build."""
    prose = "x" * 700
    parity = {"checker": prose[:80], "streams": prose[:200], "streams_checked": 32, "chunks_checked": 8192, "parity_max_abs_dp": 1.8477439880371094e-06,
              "tolerance": 1e-4, "max_prob": 1.0, "final_state_max_rel_err": 3.934e-06, "ok": True}
    roof = {"bound": "mfma", "kernel": "front_f43_kernel<32, float>", "dtype": "f32", "achieved": 106.016, "peak": 157.3, "unit": "TFLOP/s",
            "frac": 0.674, "flop_per_launch": 463856467968, "avg_launch_ms": 4.3754, "definition": prose[:130], "traffic": 5620042741,
            "traffic_profiled": {"bytes": 1, "fetch_bytes": 1, "write_bytes": 1, "source": prose[:40], "note": prose[:150]},
            "kernel_io_bytes": 4294967296, "useful_dense": {"flop_per_launch": 1, "tflops": 1.0, "note": prose[:160]},
            "issue_pipe": {"mfma_cycles_per_tile": 1, "valu_cycles_per_tile": 1, "floor_ms_at_2.4GHz": 3.7, "frac": 0.83, "mfma_share_of_floor": 0.8,
                           "definition": prose[:110]},
            "hbm": {"achieved": 981.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.12, "note": prose[:90]},
            "path": {"algorithmic_bytes": 2151677952, "algorithmic_bytes_per_chunk": 2052, "traffic": 7807203631, "traffic_profiled": 7807203631,
                     "traffic_over_algorithmic": 3.628, "traffic_profiled_over_algorithmic": 3.6, "traffic_profiled_detail": {"front": {}, "rec": {}, "note": prose[:150]},
                     "traffic_detail": {"front": {"bytes": 1}, "rec": {"bytes": 1}}, "mfma_flop_per_chunk": 573440},
            "rec_kernel": {"kernel": "rec_kernel", "avg_launch_ms": 1.13, "mfma_frac": 0.77, "single_pipe_floor_note": prose[:200], "gx_read_GBps": 1899.0,
                           "hbm_frac": 0.2374},
            "traffic_detail": {"bytes": 1, "fetch_bytes": 1, "write_bytes": 1, "unit": "bytes per launch", "how": prose[:300], "seconds": 20.0},
            "traffic_over_kernel_io": 1.309}

    def leg(name, **kw):
        d = {"value": 190049747.8, "unit": "chunks/s", "steps": 2000, "ms_per_step": 0.0813, "dtype": "i16", "n_gpus": world, "realtime_factor": 3225260.0,
             "kernel_ms": {"front": 0.05, "rec": 0.005, "note": prose[:200]}, "parity": dict(parity), "workload": prose, "sharding": prose[:90],
             "tick_latency_ms": {"median": 0.17, "p95": 0.18, "max": 0.19, "budget_ms": 32.0, "what": prose[:100]},
             "pcie": {"h2d_GBps_plain_copy": 57.3, "int16_ceiling_chunks_per_s": 5.6e7, "fraction_of_pcie_ceiling": 0.9, "bytes_per_tick": 8388608},
             "roofline": {k: roof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")},
             "sustained": {"depth": 2, "untimed_depth_trials": {"depth2": {"ticks": 300, "wall_ms": 1.0}, "depth3": {"ticks": 300, "wall_ms": 1.0}}}}
        d.update(kw)
        return d
    route = {"scheduler": "buckets", "source": "pinned", "upload": "window", "value": 5.2e7, "wall_s": 20.1, "fraction_of_pcie_ceiling": 0.946,
             "host_ms": {k: 1.0 for k in ("setup", "reserve", "slot_wait", "slot_alloc", "result_wait")}, "pcie_ceiling_chunks_per_s": 5.5e7}
    plumbing = {"workload": prose[:200], "per_call_latency_ms": {"eager_model_call_item": 0.0586, "hipgraph_step_item": 0.04, "note": prose[:150]},
                "get_speech_timestamps_60s": {"segments": 19, "one_call_fast_path_ms": 2.6, "per_chunk_protocol_ms": 116.1, "chunks": 1875,
                                              "per_chunk_protocol_ms_per_chunk": 0.06, "identical_segments": True, "cpu_beside_it": prose[:150]},
                "reference_claim": prose[:120], "config": {"workload": "configs[0]"}}
    out = {"metric": "audio-chunks/sec (32 ms @ 16 kHz)", "value": 1520397982.4, "unit": "chunks/s", "n_gpus": world, "steps": 200, "warmup": 3,
           "ms_per_step": 5.5174, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": prose[:120],
           "config": {"workload": prose[:400], "streams_per_gpu": 4096, "chunks_per_stream": 256, "sample_rate": 16000, "sharding": prose[:40]},
           "realtime_factor": 6081591.9, "outputs_finite": True, "parity": parity, "parity_max_abs_dp": 1.8e-6, "clock_ramp_steps": 37,
           "timed_region_s": 1.1035, "path_fraction": {"dense_flop_vs_fp32_peak": 1.64, "algorithmic_bytes_vs_hbm_peak": 0.0487, "note": prose[:100]},
           "kernel_ms": {"front": 4.3754, "rec": 1.1309}, "roofline": roof,
           "other_arithmetic": {k: {"what": prose[:250], "value": 2e8, "unit": "chunks/s", "max_abs_prob_diff_vs_main": 1.0281801223754883e-06}
                                for k in ("rec_bf16x9", "all_bf16x9")},
           "other_configs": {"8k": leg("8k"), "stream": leg("stream"), "stream_host": leg("stream_host"), "stream_gaps": leg("stream_gaps", gaps={"missed_fraction": 0.1, "max_burst": 5, "what": prose[:150]}),
                             "corpus": leg("corpus", parity=None, parity_sample={"recordings_checked": 15, "one_in": 10000, "parity_sample_max_abs_dp": 2.2e-6},
                                           parity_sample_max_abs_dp=2.2202730178833008e-06,
                                           legs={k: dict(route) for k in ("main", "pinned_gather", "pinned_dma", "pageable_staged", "pinned_refill_gather")}),
                             "corpus_48k": leg("corpus_48k", legs={"main": dict(route)}),
                             "stream_8k": leg("stream_8k"), "stream_host_8k": {"error": "RuntimeError: " + prose[:280]},
                             "plumbing": plumbing, "plumbing_8k": dict(plumbing)},
           "per_rank": [{"rank": r, "local_rank": r, "host_threads": 2, "numa_node_bound": r // 4, "torch_pinned_peak_bytes": 10 << 30,
                         "native_pinned_bytes": 1 << 25, "device_peak_bytes_torch": 11 << 30, "max_rss_mb": 14000} for r in range(world)],
           "node_totals": {"pinned_bytes": 87 << 30, "host_threads": 16, "device_peak_bytes_torch": 94 << 30},
           "cpu_baseline": {"value": 1289545.6, "unit": "chunks/s", "cores": 16, "affinity_cpus": 256, "kind": "port", "port": "aten-operators",
                            "best_protocol": "R4_nproc_procs_1thread", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "torch": "2.10.0+rocm7.0",
                            "runs": {f"R{i}_{'protocol_name_' * 2}": {"chunks_per_s": 1289545.6123, "B": 4096, "T": 41, "threads": 16, "what": prose[:150]}
                                     for i in range(1, 6)}, "sample": prose[:460]}}
    return out


@pytest.mark.parametrize("world", [1, 8])
def test_bench_line_fits_the_driver_tail(world, tmp_path, monkeypatch, capsys):
    """the driver keeps 8 KB of stdout and round 5's 21.7 KB line did not parse.  The REAL printing function
    (bench.emit) is given a worst-case record (nine legs with their prose, 8 ranks, an error leg): stdout is ONE line of at most 8192
    bytes that strict json.loads accepts and that carries the keys the contract names; the full record goes to the detail file (stderr names it, nothing more)."""
    import json
    import bench
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    out = _worst_case_bench_record(world)
    assert len(json.dumps(out)) > 20_000                          # (the record really is of the size that broke the parser)
    out["legs"] = bench.compact_legs(out)
    bench.emit(out)
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0].encode()) <= 8192
    assert len(cap.out.encode()) <= 8192

    def strict(c):
        raise ValueError(c)                                       # NaN / Infinity are not JSON
    d = json.loads(lines[0], parse_constant=strict)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "parity", "roofline", "cpu_baseline", "legs"):
        assert k in d, k
    assert d["n_gpus"] == world and "workload" in d["config"] and not any(k.startswith("model") for k in d["config"])
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "kernel_io_bytes", "path"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "best_protocol", "runs"):
        assert k in d["cpu_baseline"], k
    assert set(d["cpu_baseline"]["runs"]) == {"R1", "R2", "R3", "R4", "R5"} and all(isinstance(v, float) for v in d["cpu_baseline"]["runs"].values())
    for k in ("max_abs_dp", "max_prob", "final_state", "ok"):
        assert k in d["parity"], k
    assert set(d["legs"]) == {"c2"} | set(out["other_configs"])
    assert d["legs"]["corpus"]["routes_of_link"]["pinned_refill_gather"] == 0.946 and "error" in d["legs"]["stream_host_8k"]
    assert "per_rank" not in d and "node_totals" in d
    assert "shed" not in d                                        # nothing had to be dropped to fit

    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [0])
        return 0
    assert longest(d) <= 150                                      # the driver's parser clips strings there
    full = json.loads((tmp_path / d["detail"]).read_text())
    assert full["per_rank"] == out["per_rank"] and full["roofline"]["traffic_detail"] == out["roofline"]["traffic_detail"]
    # stderr only NAMES the detail file: the driver's record is a tail of "stdout, then stderr", so stdout + stderr together stay
    # inside the 8 KB it keeps (a 27 KB dump on stderr would push the line out of that tail)
    assert "bench detail: " in cap.err and len(cap.err.encode()) < 200
    assert len(cap.out.encode()) + len(cap.err.encode()) <= 8192


def test_bench_line_sheds_before_it_overflows(monkeypatch, tmp_path):
    """The safety valve: a record that would not fit even in compact form (a hundred legs) still prints a parseable line under the cap,
    and says what it dropped."""
    import json
    import bench
    out = _worst_case_bench_record(1)
    out["other_configs"].update({f"leg{i}": dict(out["other_configs"]["stream"]) for i in range(100)})
    out["legs"] = bench.compact_legs(out)
    line = bench.compact_line(out)
    assert len(json.dumps(line)) <= bench.LINE_CAP and "legs" in line["shed"] and "roofline" in line and "cpu_baseline" in line



def test_scratch_allocation_is_retried_once_after_torch_returned_its_cache(monkeypatch):
    """The library allocates its scratch outside torch's caching allocator: a VAD_ERR_ALLOC (returned before anything is launched) makes
    the host side hand torch's freed blocks back to the driver and call once more -- once; other statuses and a second failure pass
    through.  (Found by the eight-ranks-on-one-GPU rehearsal: 8 x the corpus legs' cached window buffers left no room for a lane's gx.)"""
    import torch
    from silero_vad_amd import _lib
    from silero_vad_amd.engine import Engine
    log = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: log.append("sync"))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: log.append("empty"))
    answers = iter([_lib.VAD_ERR_ALLOC, _lib.VAD_OK])
    assert Engine._retry_alloc(None, lambda: next(answers)) == _lib.VAD_OK and log == ["sync", "empty"]
    answers = iter([_lib.VAD_ERR_ALLOC, _lib.VAD_ERR_ALLOC, _lib.VAD_OK])
    assert Engine._retry_alloc(None, lambda: next(answers)) == _lib.VAD_ERR_ALLOC and len(log) == 4
    answers = iter([5, _lib.VAD_OK])
    assert Engine._retry_alloc(None, lambda: next(answers)) == 5 and len(log) == 4
    assert Engine._retry_alloc(None, lambda: _lib.VAD_OK) == _lib.VAD_OK and len(log) == 4
