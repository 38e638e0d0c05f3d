"""pqa2_amd -- MI355X-native VMAF scoring path behind PQA2's VMAFAnalyzer.analyze_videos().

    from pqa2_amd import VMAFAnalyzer            # drop-in for app/vmaf_analyzer.py:18
    from pqa2_amd.engine import FeatureEngine    # one pqa_ctx (include/pqa_vmaf.h) per GPU
"""
__version__ = "0.2.0"


def __getattr__(name):  # lazy: importing the package must not need the HIP extension or torch
    if name == "VMAFAnalyzer":
        from .vmaf_analyzer import VMAFAnalyzer
        return VMAFAnalyzer
    raise AttributeError(name)
