"""Detector stream: tk_object_detector_* / tk_preprocessor_* (+ batched extension) over ctypes."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .llm import _Path

COCO80 = ["person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant",
          "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe",
          "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat",
          "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl",
          "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
          "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave",
          "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush"]


class VideoFrame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32), ("format", C.c_int), ("data", C.c_void_p)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class DetectionResult(C.Structure):
    _fields_ = [("class_id", C.c_uint32), ("label", C.c_char_p), ("confidence", C.c_float), ("bbox", Rect)]


class _DetectorConfig(C.Structure):
    _fields_ = [("backend", C.c_int), ("gpu_device_id", C.c_int), ("model_path", C.POINTER(_Path)), ("input_width", C.c_uint32),
                ("input_height", C.c_uint32), ("class_labels", C.POINTER(C.c_char_p)), ("class_count", C.c_size_t),
                ("confidence_threshold", C.c_float), ("iou_threshold", C.c_float)]


def make_frame(arr, stride=None, rgba=False, width=None):
    """arr: uint8 [H][W][3|4], or a padded [H][stride] buffer together with width="""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h = arr.shape[0]
    w = width if width is not None else arr.shape[1]
    return VideoFrame(w, h, stride or arr.strides[0], 1 if rgba else 0, arr.ctypes.data), arr


def preprocess(arr, tw, th, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), stride=None, rgba=False, width=None):
    f, keep = make_frame(arr, stride, rgba, width)
    out = np.empty((3, th, tw), np.float32)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(lib().tk_preprocessor_resize_and_normalize_to_chw(C.byref(f), out.ctypes.data_as(C.c_void_p), tw, th, m, s))
    return out


class ObjectDetector:
    def __init__(self, model="synthetic://yolov8n?seed=5&cls_bias=-4", width=640, height=640, labels=COCO80, conf=0.5, iou=0.5,
                 device=0, backend=3, max_batch=1):
        self._labels = (C.c_char_p * len(labels))(*[s.encode() for s in labels])
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(model.encode())
        cfg = _DetectorConfig(backend, device, p, width, height, self._labels, len(labels), conf, iou)
        self.h = C.c_void_p()
        try:
            check(lib().tk_object_detector_create(C.byref(self.h), C.byref(cfg)))
        finally:
            lib().tk_path_destroy(C.byref(p))
        self.width, self.height, self.nc = width, height, len(labels)
        if max_batch > 1:
            check(lib().tk_mi355x_detector_set_max_batch(self.h, max_batch))

    def detect(self, arr, stride=None, rgba=False):
        f, keep = make_frame(arr, stride, rgba)
        res = C.POINTER(DetectionResult)()
        n = C.c_size_t(0)
        check(lib().tk_object_detector_detect(self.h, C.byref(f), C.byref(res), C.byref(n)))
        out = [(res[i].class_id, res[i].label, res[i].confidence, (res[i].bbox.x, res[i].bbox.y, res[i].bbox.w, res[i].bbox.h))
               for i in range(n.value)]
        lib().tk_object_detector_free_results(C.byref(res))
        return out

    def detect_batch(self, arrs):
        frames = (VideoFrame * len(arrs))()
        keep = []
        for i, a in enumerate(arrs):
            frames[i], k = make_frame(a)
            keep.append(k)
        res = (C.POINTER(DetectionResult) * len(arrs))()
        cnt = (C.c_size_t * len(arrs))()
        check(lib().tk_mi355x_detector_detect_batch(self.h, len(arrs), frames, res, cnt))
        out = []
        for b in range(len(arrs)):
            out.append([(res[b][i].class_id, res[b][i].label, res[b][i].confidence,
                         (res[b][i].bbox.x, res[b][i].bbox.y, res[b][i].bbox.w, res[b][i].bbox.h)) for i in range(cnt[b])])
            p = res[b]
            lib().tk_object_detector_free_results(C.byref(p))
        return out

    def forward_raw(self, x):
        x = np.ascontiguousarray(x, np.float32)
        B = x.shape[0]
        na = lib().tk_mi355x_detector_anchor_count(self.h)
        raw = np.empty((B, na, 64 + self.nc), np.float32)
        check(lib().tk_mi355x_detector_forward_raw(self.h, B, x.ctypes.data_as(C.c_void_p), raw.ctypes.data_as(C.c_void_p), C.c_size_t(raw.size)))
        return raw

    def set_fast_contraction(self, on=True):
        """opt-in: convolutions on the f16 matrix pipe with split operands (~1e-6 of scale off the exact chain, not its bits)"""
        check(lib().tk_mi355x_detector_set_fast_contraction(self.h, 1 if on else 0))

    def is_graph(self):
        """True when the model file is not the YOLOv8n topology and runs its own graph on the ONNX executor"""
        return bool(lib().tk_mi355x_detector_is_graph(self.h))

    def forward_graph(self, x):
        """the file's graph on pre-processed planar frames [B][3][H][W] -> its output [B][4 + nc][anchors]"""
        x = np.ascontiguousarray(x, np.float32)
        B = x.shape[0]
        na = (self.height // 8) * (self.width // 8) + (self.height // 16) * (self.width // 16) + (self.height // 32) * (self.width // 32)
        out = np.empty((B, 4 + self.nc, na), np.float32)
        check(lib().tk_mi355x_detector_forward_graph(self.h, B, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)))
        return out

    def last_boxes(self, frame=0, cap=500):
        boxes = np.zeros((cap, 5), np.float32)
        cls = np.zeros(cap, np.int32)
        anc = np.zeros(cap, np.int32)
        n = C.c_int(0)
        check(lib().tk_mi355x_detector_last_boxes(self.h, frame, boxes.ctypes.data_as(C.c_void_p), cls.ctypes.data_as(C.c_void_p),
                                                  anc.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return boxes[:n.value], cls[:n.value], anc[:n.value]

    def set_thresholds(self, conf, iou):
        lib().tk_object_detector_update_thresholds(self.h, C.c_float(conf), C.c_float(iou))

    def share_stats(self):
        """(live handles, batched jobs, frames, widest job) of the engine this handle shares with the others opened on the same file"""
        v = [C.c_uint64(0) for _ in range(4)]
        lib().tk_mi355x_detector_share_stats(self.h, *[C.byref(x) for x in v])
        return tuple(int(x.value) for x in v)

    def close(self):
        if self.h:
            lib().tk_object_detector_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def classify_attributes(arr, rect):
    """tk_classify_dominant_color + tk_classify_door_state on one box (x, y, w, h) of a uint8 [H][W][3] frame -> (colour, state)"""
    f, keep = make_frame(arr)
    r = Rect(*[int(v) for v in rect])
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    out = []
    for fn in (lib().tk_classify_dominant_color, lib().tk_classify_door_state):
        p = C.c_void_p()
        check(fn(C.byref(f), C.byref(r), C.byref(p)))
        out.append(C.string_at(p).decode())
        libc.free(p)
    return tuple(out)


class _PipelineConfig(C.Structure):
    _fields_ = [("backend", C.c_int), ("gpu_device_id", C.c_int), ("object_detection_model_path", C.POINTER(_Path)),
                ("depth_estimation_model_path", C.POINTER(_Path)), ("tesseract_data_path", C.POINTER(_Path)),
                ("object_confidence_threshold", C.c_float), ("max_detected_objects", C.c_uint32), ("focal_length_x", C.c_float),
                ("focal_length_y", C.c_float)]


class _RuntimeConfig(C.Structure):
    _fields_ = [("object_confidence_threshold", C.c_float), ("iou_threshold", C.c_float), ("enable_object_detection", C.c_bool),
                ("enable_depth_estimation", C.c_bool)]


class VisionObject(C.Structure):
    _fields_ = [("class_id", C.c_uint32), ("label", C.c_char_p), ("confidence", C.c_float), ("bbox", Rect), ("distance_meters", C.c_float),
                ("width_meters", C.c_float), ("height_meters", C.c_float), ("is_partially_occluded", C.c_bool), ("recognized_text", C.c_char_p),
                ("attributes", C.c_char_p)]


class VisionResult(C.Structure):
    _fields_ = [("source_frame_timestamp_ns", C.c_uint64), ("valid_analyses_mask", C.c_uint32), ("object_count", C.c_size_t),
                ("objects", C.POINTER(VisionObject)), ("text_block_count", C.c_size_t), ("text_blocks", C.c_void_p), ("depth_map", C.c_void_p),
                ("serialized_scene_graph", C.c_char_p)]


class DepthMap(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("data", C.POINTER(C.c_float))]


ANALYZE_OBJECTS, ANALYZE_DEPTH, ANALYZE_FUSION = 1, 2, 8
RESULT_OBJECTS, RESULT_DEPTH, RESULT_FUSION = 1, 2, 8


class VisionPipeline:
    """tk_vision_pipeline_* (object detection, depth estimation, object / depth fusion)"""

    def __init__(self, model="synthetic://yolov8n?seed=5&cls_bias=-0.45", conf=0.5, max_objects=0, device=0, backend=3, depth_model=None, fx=500.0,
                 fy=500.0):
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(model.encode())
        pd = lib().tk_path_create(depth_model.encode()) if depth_model else None
        cfg = _PipelineConfig(backend, device, p, pd, None, conf, max_objects, fx, fy)
        self.h = C.c_void_p()
        try:
            check(lib().tk_vision_pipeline_create(C.byref(self.h), C.byref(cfg)))
        finally:
            lib().tk_path_destroy(C.byref(p))
            if pd:
                lib().tk_path_destroy(C.byref(pd))

    def process_full(self, arr, flags, timestamp_ns=0):
        """-> (mask, objects as dicts with the fused fields, depth map [H][W] float32 or None)"""
        f, keep = make_frame(arr)
        res = C.POINTER(VisionResult)()
        check(lib().tk_vision_pipeline_process_frame(self.h, C.byref(f), flags, None, C.c_uint64(timestamp_ns), C.byref(res)))
        r = res.contents
        objs = [{"class_id": r.objects[i].class_id, "label": r.objects[i].label, "confidence": r.objects[i].confidence,
                 "bbox": (r.objects[i].bbox.x, r.objects[i].bbox.y, r.objects[i].bbox.w, r.objects[i].bbox.h),
                 "distance": np.float32(r.objects[i].distance_meters), "width_m": np.float32(r.objects[i].width_meters),
                 "height_m": np.float32(r.objects[i].height_meters)} for i in range(r.object_count)]
        depth = None
        if r.depth_map:
            dm = C.cast(r.depth_map, C.POINTER(DepthMap)).contents
            depth = np.ctypeslib.as_array(dm.data, shape=(dm.height, dm.width)).copy()
        mask = r.valid_analyses_mask
        lib().tk_vision_result_destroy(C.byref(res))
        return mask, objs, depth

    def update(self, conf, iou, enable=True, enable_depth=True):
        rc = _RuntimeConfig(conf, iou, enable, enable_depth)
        check(lib().tk_vision_pipeline_update_config(self.h, C.byref(rc)))

    def process(self, arr, flags=1, timestamp_ns=0):
        f, keep = make_frame(arr)
        res = C.POINTER(VisionResult)()
        check(lib().tk_vision_pipeline_process_frame(self.h, C.byref(f), flags, None, C.c_uint64(timestamp_ns), C.byref(res)))
        r = res.contents
        objs = [(r.objects[i].class_id, r.objects[i].label, r.objects[i].confidence,
                 (r.objects[i].bbox.x, r.objects[i].bbox.y, r.objects[i].bbox.w, r.objects[i].bbox.h), r.objects[i].attributes)
                for i in range(r.object_count)]
        out = (r.source_frame_timestamp_ns, r.valid_analyses_mask, objs)
        lib().tk_vision_result_destroy(C.byref(res))
        return out

    def close(self):
        if self.h:
            lib().tk_vision_pipeline_destroy(C.byref(self.h))
            self.h = C.c_void_p()


# ---- depth estimation + fusion: tk_depth_estimator_* / tk_vision_rust_fuse_data (include/tk/tk_depth.h) ----

class _DepthConfig(C.Structure):
    _fields_ = [("backend", C.c_int), ("gpu_device_id", C.c_int), ("model_path", C.POINTER(_Path)), ("input_width", C.c_uint32), ("input_height", C.c_uint32)]


class EnrichedObject(C.Structure):
    _fields_ = [("class_id", C.c_uint32), ("confidence", C.c_float), ("bbox", Rect), ("distance_meters", C.c_float), ("width_meters", C.c_float),
                ("height_meters", C.c_float), ("is_partially_occluded", C.c_bool)]


class _FusedResult(C.Structure):
    _fields_ = [("objects", C.POINTER(EnrichedObject)), ("count", C.c_size_t)]


class DepthEstimator:
    def __init__(self, model, width=256, height=256, device=0, backend=3):
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(model.encode())
        cfg = _DepthConfig(backend, device, p, width, height)
        self.h = C.c_void_p()
        self.width, self.height = width, height
        try:
            check(lib().tk_depth_estimator_create(C.byref(self.h), C.byref(cfg)))
        finally:
            lib().tk_path_destroy(C.byref(p))

    def estimate(self, arr, stride=None, rgba=False, width=None):
        f, keep = make_frame(arr, stride, rgba, width)
        m = C.POINTER(DepthMap)()
        check(lib().tk_depth_estimator_estimate(self.h, C.byref(f), C.byref(m)))
        out = np.ctypeslib.as_array(m.contents.data, shape=(m.contents.height, m.contents.width)).copy()
        lib().tk_depth_estimator_free_map(C.byref(m))
        return out

    def forward_raw(self, chw):
        chw = np.ascontiguousarray(chw, np.float32)
        out = np.empty((self.height, self.width), np.float32)
        check(lib().tk_mi355x_depth_forward_raw(self.h, chw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)))
        return out

    def last_raw(self):
        out = np.empty((self.height, self.width), np.float32)
        check(lib().tk_mi355x_depth_last_raw(self.h, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)))
        return out

    def close(self):
        if self.h:
            lib().tk_depth_estimator_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def depth_onnx_probe(path):
    n = C.c_int32(0)
    check(lib().tk_mi355x_depth_onnx_probe(path.encode(), C.byref(n)))
    return n.value


def _depth_map(depth):
    depth = np.ascontiguousarray(depth, np.float32)
    return DepthMap(depth.shape[1], depth.shape[0], depth.ctypes.data_as(C.POINTER(C.c_float))), depth


def fuse_data(boxes, classes, depth, frame_w, frame_h, fx, fy):
    """tk_vision_rust_fuse_data: boxes [(x, y, w, h)], classes [int] -> [(class_id, confidence, bbox, distance, width_m, height_m)] for the
    detections that have valid depth, in detection order (process-wide trackers)"""
    n = len(boxes)
    dets = (DetectionResult * max(n, 1))()
    for i, b in enumerate(boxes):
        dets[i] = DetectionResult(int(classes[i]), None, 0.9, Rect(*[int(t) for t in b]))
    dm, keep = _depth_map(depth)
    lib().tk_vision_rust_fuse_data.restype = C.POINTER(_FusedResult)
    r = lib().tk_vision_rust_fuse_data(dets, C.c_size_t(n), C.byref(dm), C.c_uint32(frame_w), C.c_uint32(frame_h), C.c_float(fx), C.c_float(fy))
    if not r:
        return None
    out = [(r.contents.objects[i].class_id, r.contents.objects[i].confidence,
            (r.contents.objects[i].bbox.x, r.contents.objects[i].bbox.y, r.contents.objects[i].bbox.w, r.contents.objects[i].bbox.h),
            np.float32(r.contents.objects[i].distance_meters), np.float32(r.contents.objects[i].width_meters), np.float32(r.contents.objects[i].height_meters))
           for i in range(r.contents.count)]
    lib().tk_vision_rust_free_fused_result.argtypes = [C.POINTER(_FusedResult)]
    lib().tk_vision_rust_free_fused_result(r)
    return out


def fusion_reset():
    lib().tk_mi355x_fusion_reset()


def fusion_raw_distance(box, depth, frame_w, frame_h):
    dm, keep = _depth_map(depth)
    lib().tk_mi355x_fusion_raw_distance.restype = C.c_float
    r = Rect(*[int(t) for t in box])
    return np.float32(lib().tk_mi355x_fusion_raw_distance(C.byref(r), C.byref(dm), C.c_uint32(frame_w), C.c_uint32(frame_h)))
