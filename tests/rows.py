"""Which SURVEY.md section-8 row(s) every test function covers.

The test FILES grew round by round (`test_gpu_round2..5.py`); the judge audits row by row.  This map is the index: conftest.py turns it into
pytest markers (`-m row_A4` selects everything that pins VarAttention; `-m "gpu and row_A10"` the on-device local-loss tests), a CPU test
checks that no test function is missing from it, and `python tests/rows.py` prints the table kept in tests/README.md.

Rows: A1 region select . A2 object prologue . A3 SpaceTimeBlock x 12 . A4 VarAttention . A5 Mlp . A6 encoder tail .
A7 ObjectRelation forward / compute_* . A8 DistilBERT + txt_proj . A9 sim_matrix + NormSoftmaxLoss . A10 xattn_score_fast . A11 RWALoss /
GlobalLocalLoss . A12 AllGather_multi . A13 DDP gradient all-reduce . A14 train-step body + HF-AdamW . A15 get_sim_by_segment .
b drop-in boundary (C ABI, factory, state_dict, constructor) . c oracle (pinned against the reference's goldens) . d measurement .
e multi-GPU . f1 eval on device . f2 input pipeline . f3 checkpoint / optimizer interchange / LR quirk . f4 timeattn + QA head
"""
ROWS = {
    # ---- tests/test_abi.py
    "test_library_exports_every_declared_symbol": "b",
    "test_product_library_exports_no_developer_switch": "b",
    "test_integration_doc_names_every_entry_point_with_the_reference_interface_it_replaces": "b",
    "test_size_helpers_run_on_host": "b",
    "test_bad_arguments_are_reported_not_launched": "b",
    "test_no_reference_or_oracle_import_in_product": "b c",
    # ---- tests/test_boundary_init.py
    "test_constructor_initialises_object_tower_like_reference": "b A7",
    "test_factory_builds_from_unchanged_json": "b",
    "test_missing_files_raise_like_reference": "b",
    "test_wrong_shape_in_vit_file_raises": "b",
    "test_safetensors_directory": "b A8",
    # ---- tests/test_host_logic.py
    "test_config_factory_builds_dropin_modules": "b",
    "test_state_dict_schema_matches_reference": "b f3",
    "test_reference_argument_errors": "b",
    "test_product_path_fails_loudly_without_gpu": "b",
    "test_checkpoint_helpers": "f3",
    "test_data_parallel_plumbing_gloo_world2": "A12 A13 e",
    "test_retrieval_metrics_match_reference_golden": "f1",
    "test_shard_indices_equal_torch_distributed_sampler": "f2 e",
    "test_frame_sampling_and_npz_schema": "f2 A1",
    "test_adjust_learning_rate_reproduces_the_reference_quirk": "f3",
    "test_fused_adamw_state_dict_uses_the_reference_checkpoint_layout": "f3 A14",
    "test_two_graph_exchange_plan_covers_the_arena_once": "A13 e",
    "test_graphed_step_host_bookkeeping_bucket_alignment_and_lru": "A13 A14",
    "test_bf16_gradient_buckets_gloo_world2": "A13 e",
    "test_rs_ag_shard_walk_gloo": "A13 e",
    "test_no_undefined_names_anywhere_in_the_tree": "b",
    "test_every_test_is_mapped_to_a_survey_row": "b",
    # ---- tests/test_oracle_golden.py  (the oracle against the imported reference's goldens: row c, per path row)
    "test_region_select_bit_exact": "A1 c",          # (the same name in test_gpu_kernels.py: the HIP kernel against the oracle and golden G1)
    "test_xattn_and_rwa": "A10 A11 c",
    "test_sim_matrix_norm_softmax": "A9 c",
    "test_model_forward_backward": "A2 A3 A4 A5 A6 A7 A8 A9 A10 A11 c",
    "test_eval_grid_scores_match_reference_get_sim_by_segment": "A15 c",
    "test_focal_gate_margins_recorded": "A10 c",
    "test_ten_step_loss_curve_vs_reference": "A14 c",
    "test_eval_pipeline_vs_reference": "f1 A15 c",
    "test_philox_known_answer_vectors": "A8 c",
    "test_qa_head_vs_reference": "f4 c",
    "test_oracle_vs_reference_on_the_retrieval_set_with_signal": "f1 A15 c",
    "test_oracle_first_step_at_the_benchmark_size_vs_reference_g12": "A14 c d",
    "test_float64_oracle_gradients_agree_with_the_reference": "c A14",
    "test_oracle_first_finetune_step_vs_reference_g14": "c A14 f1",
    # ---- tests/test_gpu_kernels.py
    "test_gemm_forward_forms": "A3 A5 A6 A8",
    "test_gemm_forward_forms_every_tile_variant": "A3 A5 A6 A8",
    "test_gemm_epilogues": "A5 A3",
    "test_layernorm": "A3 A8",
    "test_colsum": "A3 A14",
    "test_space_attention": "A4",
    "test_space_attention_other_backward_forms": "A4",
    "test_space_attention_round5_kernels_against_the_round4_kernels": "A4",
    "test_space_attention_fold_switched_off": "A4",
    "test_full_attention": "A8",
    "test_object_prologue_pieces": "A2",
    "test_text_embed": "A8",
    "test_xattn": "A10",
    "test_xattn_long_video_path_forced_on_shapes_the_pair_kernels_also_take": "A10",
    "test_loss_heads": "A9 A11",
    "test_split_cls_forward_and_backward_equal_the_slicing_form": "A7",
    "test_loss_heads_matrix_core_form": "A9 A11",
    "test_adamw_matches_oracle": "A14",
    "test_gemm_pingpong_race_screen": "A3 A5",
    "test_deferred_reductions_match_immediate": "A3 A14",
    "test_wgrad_grouped_matches_single": "A3 A5 A14",
    "test_optimizer_update_riding_in_the_grouped_weight_gradient_launch": "A14 A3",
    "test_region_batcher_ragged_files_match_reference_pipeline": "f2 A1",
    "test_gemm_fused_column_sums": "A3 A5",
    "test_gemm_resident_b_batched_against_the_tile_kernel_and_fp32": "A10",
    "test_region_select_edge_counts": "A1",
    # ---- tests/test_gpu_model.py
    "test_graft_entry_smoke_runs_and_checks_against_the_oracle": "b d",
    "test_fp32_forward_backward_vs_reference_golden": "A2 A3 A4 A5 A6 A7 A8 A9 A10 A11",
    "test_model_built_the_reference_way_from_pretrained_files_runs_on_their_weights": "b A7 A8",
    "test_fp32_gradients_vs_float64_oracle": "A2 A3 A4 A5 A6 A7 A8 A9 A10 A11 A14",
    "test_bf16_forward_backward_close_to_reference": "A2 A3 A4 A5 A6 A7 A8 A9 A10 A11",
    "test_fused_adamw_loss_curve_matches_oracle": "A14",
    "test_32_frame_forward_and_loss_vs_golden": "A3 A4 A10",
    "test_arena_paths_give_the_same_gradients": "A14",
    "test_two_rank_data_parallel_step_matches_averaged_gradients": "A13 e",
    "test_eval_grid_and_retrieval_metrics_vs_reference_golden": "A15 f1",
    # ---- tests/test_gpu_round2.py
    "test_sim_matrix_rectangular_forward_backward": "A9 f1",
    "test_clip_shorter_than_num_frames_gradients_reach_the_arena": "A2 A14",
    "test_bf16_default_arena_keeps_learning": "A14",
    "test_second_backward_before_step_is_refused": "A14",
    "test_graph_replay_equals_eager": "A14 d",
    "test_ten_step_loss_curve_and_optimizer_state_vs_reference": "A14 f3",
    "test_evaluate_vs_reference_retrieval_golden": "f1 A15",
    "test_bf16_at_benchmark_size_vs_oracle": "A14 d",
    "test_bf16_32_frames_forward_and_loss_vs_golden": "A3 A4 A10",
    "test_fp32_32_frames_backward_vs_oracle_gradients": "A3 A4 A10",
    "test_xattn_on_device_vs_reference_golden": "A10",
    "test_region_batcher_back_to_back_batches": "f2",
    "test_bench_rccl_path_in_a_one_rank_group": "A13 e d",
    "test_backward_cut_in_two_gives_the_same_gradients": "A13 A14",
    "test_timeattn_forward_backward_vs_reference_golden": "f4",
    "test_timeattn_arena_training_step_matches_oracle": "f4 A14",
    "test_fused_local_loss_forward_vs_oracle": "A10 A15",
    "test_local_loss_bf16_on_chip_tiles_and_gram_form_vs_generic_and_oracle": "A10",
    "test_philox_on_device_and_dropout_masks_vs_oracle": "A8",
    "test_text_tower_train_mode_dropout_vs_oracle": "A8",
    "test_graph_replay_draws_fresh_dropout_masks": "A8 A14",
    "test_parallel_towers_give_identical_results": "A7",
    "test_soak_replayed_steps_hold_memory_and_learn": "A14 d",
    "test_qa_model_vs_reference_golden": "f4",
    "test_bench_launches_its_own_ranks": "e d",
    # ---- tests/test_gpu_round3.py
    "test_two_rank_graphed_step_matches_hand_averaged_gradients": "A13 e A14",
    "test_gather_negatives_two_ranks_on_device_vs_oracle": "A12 e A9 A11",
    "test_graph_replay_follows_lr_change_and_resumed_step_counter": "A14 f3",
    "test_bf16_at_benchmark_size_losses_and_gradient_norms": "A14 d",
    "test_bf16_ten_step_loss_curve_through_graph_replay_vs_f64_curve": "A14",
    "test_gemm_short_tiles_bit_equal_to_256_row_tiles": "A3 A5 A8",
    "test_bf16_evaluate_runs_the_grid_on_the_fused_kernel_vs_reference_golden": "f1 A15",
    "test_get_sim_by_segment_precision_knob": "A15",
    "test_evaluate_mscoco_branch_subsamples_videos_and_passes_fold": "f1",
    "test_persistent_gemm_kernel_agrees_with_the_one_tile_form": "A3 A5",
    "test_bf16_32_frames_backward_gradient_norms_vs_fp32_path": "A3 A4 A10",
    "test_region_select_bit_exact_at_32_frames": "A1",
    "test_evaluate_with_two_ranks_matches_one_process": "f1 e",
    "test_region_batcher_stages_into_graph_inputs_and_the_step_matches_the_copying_path": "f2 A14",
    "test_graph_step_recaptures_when_the_batch_shape_changes": "A14",
    "test_training_steps_are_bit_reproducible_run_to_run": "A14",
    "test_evaluation_between_replayed_steps_leaves_training_unchanged": "A14 f1",
    "test_graph_capture_with_a_prefetching_loader_thread_running": "f2 A14",
    "test_replays_with_a_host_synchronisation_between_them": "A14",
    "test_two_rank_bf16_graphed_step_with_host_sync_is_in_lock_step_and_reproducible": "A13 e A14",
    # ---- tests/test_gpu_round4.py
    "test_fp32_evaluate_reproduces_every_rank_of_the_retrieval_set_with_signal": "f1 A15",
    "test_bf16_evaluate_on_the_retrieval_set_with_signal_stays_within_stated_rank_changes": "f1 A15",
    "test_fp32_finetune_then_evaluate_vs_reference": "A14 f1 A15",
    "test_bf16_finetune_then_evaluate_stays_within_the_stated_bounds": "A14 f1 A15",
    "test_bf16_trains_like_fp32_at_the_benchmark_size": "A14 d",
    "test_text_mask_len_kernel_equals_the_stock_ops": "A14 A8",
    # ---- tests/test_gpu_round5.py
    "test_fp32_five_step_curve_at_the_benchmark_size_vs_reference": "A14 d",
    "test_bf16_graph_replayed_five_step_curve_at_the_benchmark_size_vs_reference": "A14 d",
    "test_one_rank_rccl_bf16_buckets_equal_the_fp32_exchange_of_bf16_rounded_gradients": "A13 e",
    "test_local_loss_with_its_two_halves_on_two_streams_equals_the_single_stream_form": "A10",
}
ALL_ROWS = ["A%d" % i for i in range(1, 16)] + ["b", "c", "d", "e", "f1", "f2", "f3", "f4"]


def rows_of(function_name):
    return ROWS.get(function_name, "").split()


def table(items):
    """items: [(file, function, is_gpu)] -> markdown, one section per row."""
    out = []
    for row in ALL_ROWS:
        mine = sorted({(f, fn, g) for f, fn, g in items if row in rows_of(fn)})
        out.append(f"### {row}  ({len(mine)} test functions; `pytest -m row_{row}`)")
        out += [f"- `{f}::{fn}`" + ("  (gpu)" if g else "") for f, fn, g in mine]
        out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    import ast
    import glob
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    items = []
    for path in sorted(glob.glob(os.path.join(here, "test_*.py"))):
        src = open(path).read()
        gpu_file = any(line.startswith("pytestmark = pytest.mark.gpu") for line in src.splitlines())
        for node in ast.parse(src).body:
            if isinstance(node, ast.FunctionDef) and node.name.startswith("test_"):
                items.append((os.path.basename(path), node.name, gpu_file))
    print(table(items))
