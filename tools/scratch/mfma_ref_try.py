import ctypes as C, time
L=C.CDLL('tools/libmfma_ref.so')
for shape in (0,1,2,3,0,1):
    tf=C.c_double(); ck=C.c_double(); n=C.c_int()
    t=time.time(); rc=L.mfma_ref_rate(shape, C.c_double(2.0), C.byref(tf), C.byref(ck), C.byref(n))
    print(shape, rc, tf.value, ck.value, n.value, time.time()-t)
