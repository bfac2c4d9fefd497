"""``process``: the public entry point (matchering/core.py:32-121) -- load, check, validate,
``stages.main`` on the GPU, save, previews.  Same signature, log codes and exceptions."""

from .audio_io import load, save
from .checker import check, check_equality
from .config import Config
from .log import Code, ModuleError, debug, debug_line, info
from .preview import create_preview
from .results import Result
from .stages import main
from .utils import get_temp_folder


def process(target: str, reference: str, results: list, config: Config = None,
            preview_target: Result = None, preview_result: Result = None):
    config = config if config is not None else Config()
    debug("Please give us a star to help the project: https://github.com/sergree/matchering")
    debug_line()
    info(Code.INFO_LOADING)
    if not results:
        raise RuntimeError("The result list is empty")
    temp_folder = config.temp_folder if config.temp_folder else get_temp_folder(results)

    target, target_sample_rate = load(target, "target", temp_folder)
    target, target_sample_rate = check(target, target_sample_rate, config, "target")
    reference, reference_sample_rate = load(reference, "reference", temp_folder)
    reference, reference_sample_rate = check(reference, reference_sample_rate, config, "reference")
    if not config.allow_equality:
        check_equality(target, reference)

    if (not (target_sample_rate == reference_sample_rate == config.internal_sample_rate)
            or not (target.shape[1] == reference.shape[1] == 2)
            or not (target.shape[0] > config.fft_size and reference.shape[0] > config.fft_size)):
        raise ModuleError(Code.ERROR_VALIDATION)

    result, result_no_limiter, result_no_limiter_normalized = main(
        target, reference, config,
        need_default=any(rr.use_limiter for rr in results),
        need_no_limiter=any(not rr.use_limiter and not rr.normalize for rr in results),
        need_no_limiter_normalized=any(not rr.use_limiter and rr.normalize for rr in results))
    del reference

    debug_line()
    info(Code.INFO_EXPORTING)
    for wanted in results:
        if wanted.use_limiter:
            chosen = result
        else:
            chosen = result_no_limiter_normalized if wanted.normalize else result_no_limiter
        save(wanted.file, chosen, config.internal_sample_rate, wanted.subtype)

    if preview_target or preview_result:
        shown = next(item for item in (result, result_no_limiter, result_no_limiter_normalized) if item is not None)
        create_preview(target, shown, config, preview_target, preview_result)

    debug_line()
    info(Code.INFO_COMPLETED)
