"""The reference's training.py loop (training.py:43-307) on the MI355X path with the synthetic stand-in loader: loader ->
shape-keyed step table -> train_step -> loss CSV -> per-chunk save_model (+ -EMA) -> training-state file for resume.
One process per GPU:  python -m torch.distributed.run --nproc-per-node N examples/train_synthetic.py config.json
(single GPU: python examples/train_synthetic.py config.json).  config.json holds the reference's model_properties.json keys
(model_properties_example.json) plus "batches_per_chunk"; "model_path" is a diffusers-Flax pipeline directory."""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from stable_diffusion_training_amd import dp
from stable_diffusion_training_amd import training_utils as tu
from stable_diffusion_training_amd.checkpoint import gather_rng_states
from stable_diffusion_training_amd.streamer import DataLoader


def delete_file_or_folder(path):
    if os.path.isdir(path):
        shutil.rmtree(path, ignore_errors=True)
    elif os.path.exists(path):
        os.remove(path)


def main(config_dict, models=None, tokenizer=None, log=print):
    """models: optional load_models-style dict (tests pass seeded weights); otherwise read from config_dict["model_path"]."""
    assert len(config_dict["image_area_root"]) == len(config_dict["minimum_axis_length"]), \
        "number of elements in image_area_root and minimum_axis_length is not match! check your config files!"
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=dp.rccl_group_options())
    training_config = tu.TrainingConfig.from_dict(config_dict)
    if models is None:
        models = tu.load_models(training_config)
        tokenizer = models["tokenizer"]
    dataloader = DataLoader(
        tokenizer_obj=tokenizer, config=None, ramdisk_path=config_dict.get("ramdisk_path"),
        training_batch_size=config_dict["batch_size"], repeat_batch=config_dict["repeat_batch"],
        maximum_resolution_areas=[x ** 2 for x in config_dict["image_area_root"]],
        bucket_lower_bound_resolutions=config_dict["minimum_axis_length"],
        numb_of_worker_thread=config_dict.get("numb_of_dataloader_worker_thread", 1),
        queue_get_timeout=config_dict.get("queue_get_timeout", 60), chunk_number=config_dict["chunk_number"],
        seed=config_dict["master_seed"], context_concatenation_multiplier=config_dict["context_window_concatenation_count"],
        batches_per_chunk=config_dict.get("batches_per_chunk", 100), vocab_size=models["text_encoder"]["config"]["vocab_size"],
        rank=rank, world_size=world, device=dev)
    dataloader._print_debug = bool(config_dict.get("DEBUG"))

    train_rngs = torch.Generator(device=dev)
    train_rngs.manual_seed(config_dict["master_seed"] * 1009 + rank)  # different noise / timesteps on every shard
    (unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params, frozen_vae, frozen_schedulers,
     model_object_dict) = tu.on_device_model_training_state(training_config, models, device=dev)
    reducer = dp.GradReducer([unet_state.store, text_encoder_state.store]) if world > 1 else None
    train_step_funcs = tu.dp_compile_all_unique_resolution(
        unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params, frozen_vae, frozen_schedulers, training_config,
        reducer=reducer, per_device_batch=config_dict["batch_size"] // world)
    resume = config_dict.get("resume_training_state")
    if resume and os.path.exists(resume):
        tu.load_training_state(resume, unet_state, text_encoder_state, train_rngs, rank=rank, world=world)
        log(f"resumed optimizer / RNG state from {resume} at step {unet_state.step}")

    if rank == 0 and not os.path.isfile(config_dict["loss_csv"]):
        with open(config_dict["loss_csv"], "w") as f:
            f.write("steps, step_size, loss, time, chunk, seed\n")

    def save(ema):
        base = config_dict["model_path"].split("@")[0] + ("-EMA" if ema else "")
        up = unet_ema_params if (ema and config_dict["accumulate_unet_ema"]) else unet_state.params
        tp = text_encoder_ema_params if (ema and config_dict["accumulate_text_encoder_ema"]) else text_encoder_state.params
        tu.save_model(model_object_dict, tokenizer, up, tp, frozen_vae.params, f'{base}@{config_dict["chunk_steps"]}')
        delete_file_or_folder(f'{base}@{config_dict["chunk_steps"] - config_dict["keep_trained_model_buffer"]}')

    losses = []
    for _ in range(config_dict["chunk_limit"]):
        if config_dict["chunk_number"] >= config_dict["chunk_limit"]:
            config_dict["chunk_number"] = 0
        dataloader.chunk_number = config_dict["chunk_number"]
        dataloader.grab_and_prefetch_chunk(numb_of_prefetched_batch=config_dict.get("numb_of_prefetched_batch", 1))
        dataloader.prepare_training_dataframe()
        dataloader.create_training_dataframe()
        dataloader.dispatch_worker()
        if reducer is not None:
            reducer.gather_state()  # sharded optimizer: whole state on every rank before the rank-0 save (collective; no-op otherwise)
        if rank == 0:  # pre-flight save (training.py:149-184): fail before the chunk, not after it
            tu.save_model(model_object_dict, tokenizer, unet_state.params, text_encoder_state.params, frozen_vae.params,
                          config_dict["test_save_path"])
            delete_file_or_folder(config_dict["test_save_path"])
        start = time.time()
        train_metrics = []
        for count in range(int(dataloader._bulk_batch_count + dataloader._first_batch_count)):
            current_batch = dataloader.grab_next_batch()
            if current_batch == "end_of_batch":
                break
            if current_batch is None:
                continue
            w = config_dict["text_encoder_context_window"]
            current_batch["input_ids"] = current_batch["input_ids"].reshape(-1, w)
            current_batch["attention_mask"] = current_batch["attention_mask"].reshape(-1, w)
            (unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params, train_metric, train_rngs) = \
                train_step_funcs[current_batch["pixel_values"].shape](
                    unet_state, text_encoder_state, unet_ema_params, text_encoder_ema_params, current_batch, train_rngs,
                    frozen_vae, frozen_schedulers)
            train_metrics.append(train_metric["loss"])  # device scalars: reading them below is the only synchronisation
            if count % config_dict["loss_logging_interval"] == 0:
                loss = float(sum(train_metrics) / len(train_metrics))
                losses.append(loss)
                elapsed = round(time.time() - start, 4)
                start = time.time()
                train_metrics = []
                if rank == 0:
                    log(f'at steps {count}, avg loss for {config_dict["loss_logging_interval"]} steps: {loss}, took {elapsed} second(s)')
                    with open(config_dict["loss_csv"], "a") as f:
                        f.write(f'\n{count},{config_dict["loss_logging_interval"]},{loss},{elapsed},{config_dict["chunk_steps"]},{config_dict["master_seed"]}')
        rng_states = gather_rng_states(train_rngs)  # every rank resumes ITS noise / timestep stream (collective)
        if reducer is not None:
            reducer.gather_state()                  # ... and the sharded optimizer's state becomes whole (collective, all ranks)
        if rank == 0:
            save(ema=False)
            if config_dict["ema_rate"]:
                save(ema=True)
            state_path = config_dict["model_path"].split("@")[0] + "-state.safetensors"
            tu.save_training_state(state_path, unet_state, text_encoder_state, rng_states=rng_states)
        config_dict["model_path"] = f'{config_dict["model_path"].split("@")[0]}@{config_dict["chunk_steps"]}'  # training.py:301-304
        config_dict["chunk_number"] += 1
        config_dict["chunk_steps"] += 1
    if world > 1:
        dist.barrier()
    return losses, unet_state, text_encoder_state


if __name__ == "__main__":
    with open(sys.argv[1] if len(sys.argv) > 1 else "model_properties.json") as f:
        cfg = json.load(f)
    main(cfg)
