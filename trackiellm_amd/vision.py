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


class VisionPipeline:
    """tk_vision_pipeline_* (object-detection analysis)"""

    def __init__(self, model="synthetic://yolov8n?seed=5&cls_bias=-0.45", conf=0.5, max_objects=0, device=0, backend=3):
        lib().tk_path_create.restype = C.POINTER(_Path)
        p = lib().tk_path_create(model.encode())
        cfg = _PipelineConfig(backend, device, p, None, None, conf, max_objects, 500.0, 500.0)
        self.h = C.c_void_p()
        try:
            check(lib().tk_vision_pipeline_create(C.byref(self.h), C.byref(cfg)))
        finally:
            lib().tk_path_destroy(C.byref(p))

    def update(self, conf, iou, enable=True):
        rc = _RuntimeConfig(conf, iou, enable, False)
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
