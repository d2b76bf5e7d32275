{-# LANGUAGE NamedFieldPuns #-}
{-# LANGUAGE ForeignFunctionInterface #-}
-- |
-- Module      : Streamly.Internal.LZ4.GPU
--
-- Batched MI355X replacements for the two per-block primitives of
-- "Streamly.Internal.LZ4" (@compressChunk@ / @decompressChunk@) and for the two
-- combinators that call them once per array (@compressChunksD@,
-- @decompressChunksRawD@).  Everything else of the package (resizeChunksD, the
-- frame parser, Streamly.LZ4) is used unchanged.
--
-- UNTESTED SOURCE: GHC, cabal and streamly-0.8.2 are not available in the build
-- image of this repository, so this module has not been compiled.  The C++ mirror
-- (include/streamly_lz4.hpp, streamly-lz4_amd/csrc/host_stream.cpp) implements the
-- same restructuring and is what the test-suite exercises.  Link with
--
-- >   include-dirs:    <repo>/include
-- >   extra-lib-dirs:  <repo>/streamly-lz4_amd/lib
-- >   extra-libraries: mi355lz4
--
-- Design: the incoming @Array Word8@ stream is grouped into batches of up to
-- 'batchBlocks' arrays; one @ccall safe@ hands a batch to the GPU (a GPU round trip must
-- not block a capability the way the reference's @ccall unsafe@ per-block calls may).
-- Compressed blocks are INDEPENDENT by default, so no previous input has to be kept alive
-- ('setLinkedCompress' makes the blocks of one batch a linked stream like the reference's:
-- a batch is handed over as a whole, so its arrays are alive for the duration of the call);
-- decompression keeps the previous OUTPUT array alive exactly like the reference
-- (it is the dictionary of the next block of a linked stream).
module Streamly.Internal.LZ4.GPU
    ( Engine
    , newEngine
    , freeEngine
    , setLinkedCompress
    , MultiEngine
    , newMultiEngine
    , freeMultiEngine
    , c_multiCompressBatch
    , c_multiDecompressBatch
    , compressChunksGPU
    , decompressChunksRawGPU
    )
where

import Control.Monad (forM, forM_, when)
import Control.Monad.IO.Class (MonadIO(..))
import Data.Int (Int32)
import Data.Word (Word8)
import Foreign.C (CInt(..), CSize(..))
import Foreign.Marshal.Alloc (alloca)
import Foreign.Marshal.Array (allocaArray, peekArray, pokeArray)
import Foreign.Marshal.Utils (copyBytes)
import Foreign.Ptr (Ptr, nullPtr, plusPtr, castPtr)
import Foreign.Storable (peek)

import qualified Streamly.Internal.Data.Array.Foreign as Array
import qualified Streamly.Internal.Data.Array.Foreign.Type as Array
import qualified Streamly.Internal.Data.Array.Foreign.Mut.Type as MArray
import qualified Streamly.Internal.Data.Fold as Fold
import qualified Streamly.Internal.Data.Stream.StreamD as Stream

import Streamly.Internal.LZ4.Config

data C_Engine
newtype Engine = Engine (Ptr C_Engine)

foreign import ccall safe "mi355lz4.h mi355lz4_create"
    c_create :: Ptr (Ptr C_Engine) -> CInt -> IO CInt
foreign import ccall safe "mi355lz4.h mi355lz4_destroy"
    c_destroy :: Ptr C_Engine -> IO ()
foreign import ccall unsafe "mi355lz4.h mi355lz4_compress_bound"
    c_bound :: CInt -> CInt
-- small calls: segments per block (-1 automatic, 0 off, 2..64); include/mi355lz4.h
foreign import ccall unsafe "mi355lz4.h mi355lz4_set_segments"
    c_setSegments :: Ptr C_Engine -> CInt -> IO CInt

-- linked device decodes without a host wait (0 = default: wait); include/mi355lz4.h
foreign import ccall unsafe "mi355lz4.h mi355lz4_set_linked_async"
    c_setLinkedAsync :: Ptr C_Engine -> CInt -> IO CInt

foreign import ccall unsafe "mi355lz4.h mi355lz4_set_linked_compress"
    c_setLinkedCompress :: Ptr C_Engine -> CInt -> IO CInt

-- replaces c_compressFastContinue (Streamly/Internal/LZ4.hs:123-131), N blocks per call
foreign import ccall safe "mi355lz4.h mi355lz4_compress_batch"
    c_compressBatch
        :: Ptr C_Engine -> Ptr (Ptr Word8) -> Ptr Int32 -> CInt -> CInt -> CInt
        -> Ptr Word8 -> CSize -> Ptr CSize -> Ptr Int32 -> Ptr Int32 -> IO CInt

-- replaces c_decompressSafeContinue (Streamly/Internal/LZ4.hs:133-140), N blocks per call
foreign import ccall safe "mi355lz4.h mi355lz4_decompress_batch"
    c_decompressBatch
        :: Ptr C_Engine -> Ptr Word8 -> CSize -> CInt -> CInt -> CInt
        -> Ptr Word8 -> CInt -> Ptr Word8 -> CSize -> Ptr CSize -> Ptr Int32
        -> CInt -> Ptr CInt -> IO CInt

-- Many linked streams (each the output of one reference compressChunks pipeline) in one call:
-- stream s = blocks [streamFirst[s], streamFirst[s+1]).  A server that decompresses many files or
-- connections at once gathers their resized blocks and calls this instead of one
-- c_decompressBatch per stream: a single linked stream is walked by one wavefront (slow), many
-- streams run side by side.
foreign import ccall safe "mi355lz4.h mi355lz4_decompress_streams"
    c_decompressStreams
        :: Ptr C_Engine -> Ptr Word8 -> CSize -> CInt -> CInt -> Ptr Int32 -> CInt
        -> Ptr Word8 -> CSize -> Ptr CSize -> Ptr Int32 -> CInt -> Ptr CInt -> IO CInt

-- Several GPUs behind one handle (one process, host buffers: a host caller is bound by PCIe, one link per GPU).  The two
-- batch calls take the arguments of c_compressBatch / c_decompressBatch (independent blocks) and give the same bytes; the
-- batch is cut into one contiguous block range per device and the results lie in order in the caller's buffer
-- (include/mi355lz4.h, "several GPUs behind one handle").
data C_Multi
newtype MultiEngine = MultiEngine (Ptr C_Multi)

foreign import ccall safe "mi355lz4.h mi355lz4_create_multi"
    c_createMulti :: Ptr (Ptr C_Multi) -> Ptr CInt -> CInt -> IO CInt
foreign import ccall safe "mi355lz4.h mi355lz4_destroy_multi"
    c_destroyMulti :: Ptr C_Multi -> IO ()
foreign import ccall safe "mi355lz4.h mi355lz4_multi_compress_batch"
    c_multiCompressBatch
        :: Ptr C_Multi -> Ptr (Ptr Word8) -> Ptr Int32 -> CInt -> CInt -> CInt
        -> Ptr Word8 -> CSize -> Ptr CSize -> Ptr Int32 -> Ptr Int32 -> IO CInt
foreign import ccall safe "mi355lz4.h mi355lz4_multi_decompress_batch"
    c_multiDecompressBatch
        :: Ptr C_Multi -> Ptr Word8 -> CSize -> CInt -> CInt
        -> Ptr Word8 -> CSize -> Ptr CSize -> Ptr Int32 -> CInt -> Ptr CInt -> IO CInt

-- | One handle over the given HIP devices (e.g. @[0 .. 7]@ for a node).
newMultiEngine :: [Int] -> IO MultiEngine
newMultiEngine devs = alloca $ \pp -> allocaArray (length devs) $ \pd -> do
    pokeArray pd (map fromIntegral devs)
    rc <- c_createMulti pp pd (fromIntegral (length devs))
    when (rc /= 0) $ error "mi355lz4_create_multi failed (no gfx950 device?)"
    MultiEngine <$> peek pp

freeMultiEngine :: MultiEngine -> IO ()
freeMultiEngine (MultiEngine p) = c_destroyMulti p

newEngine :: Int -> IO Engine
newEngine dev = alloca $ \pp -> do
    rc <- c_create pp (fromIntegral dev)
    when (rc /= 0) $ error "mi355lz4_create failed (no gfx950 device?)"
    Engine <$> peek pp

freeEngine :: Engine -> IO ()
freeEngine (Engine p) = c_destroy p

-- | 'True': 'compressChunksGPU' writes a linked stream (the block before is a block's
-- dictionary, what @LZ4_compress_fast_continue@ does with the previous chunk,
-- Streamly/Internal/LZ4.hs:376,389); 'decompressChunksRawGPU' reads either kind.
setLinkedCompress :: Engine -> Bool -> IO ()
setLinkedCompress (Engine p) on = do
    rc <- c_setLinkedCompress p (if on then 1 else 0)
    when (rc /= 0) $ error "mi355lz4_set_linked_compress failed"

batchBlocks :: Int
batchBlocks = 4096

metaSizeOf :: BlockConfig -> Int
metaSizeOf BlockConfig {blockSize} = case blockSize of
    BlockHasSize -> 8
    _ -> 4

fixedUncompOf :: BlockConfig -> Int
fixedUncompOf BlockConfig {blockSize} = case blockSize of
    BlockHasSize -> 0
    BlockMax64KB -> 64 * 1024
    BlockMax256KB -> 256 * 1024
    BlockMax1MB -> 1024 * 1024
    BlockMax4MB -> 4 * 1024 * 1024

-- | One GPU call for a batch of arrays: the batched form of @compressChunk@.
compressBatch :: Engine -> BlockConfig -> Int -> [Array.Array Word8] -> IO [Array.Array Word8]
compressBatch (Engine eng) cfg speed arrs = do
    let n = length arrs
        meta = metaSizeOf cfg
        lens = map Array.byteLength arrs
        cap = sum (map (\l -> fromIntegral (c_bound (fromIntegral l)) + meta) lens)
    (MArray.Array cont dstBegin_ dstBegin dstMax) <- MArray.newArray (max cap 1)
    allocaArray n $ \pSrc -> allocaArray n $ \pLen -> allocaArray n $ \pFlen ->
      allocaArray n $ \pStatus -> alloca $ \pOutLen -> do
        -- pin every source for the duration of the call (Array.asPtrUnsafe nests)
        let withAll [] k = k []
            withAll (a:as) k = Array.asPtrUnsafe (Array.unsafeCast a) $ \p -> withAll as (k . (p :))
        withAll arrs $ \ptrs -> do
            pokeArray pSrc ptrs
            pokeArray pLen (map fromIntegral lens)
            rc <- c_compressBatch eng pSrc pLen (fromIntegral n) (fromIntegral speed)
                      (fromIntegral meta) dstBegin (fromIntegral cap) pOutLen pFlen pStatus
            when (rc /= 0) $ error "compressChunks: mi355lz4_compress_batch failed"
        flens <- map fromIntegral <$> peekArray n pFlen
        -- one Array per block: views into the batch buffer, as resizeChunksD does (:480-484)
        let offs = scanl (+) 0 flens
        return [ Array.unsafeFreeze (MArray.Array cont dstBegin_ (dstBegin `plusPtr` (o + l)) dstMax)
                   `seq` Array.Array cont (dstBegin `plusPtr` o) (dstBegin `plusPtr` (o + l))
               | (o, l) <- zip offs flens ]

-- | One freshly allocated array holding the given arrays back to back (the reference splices
-- pairwise with @Array.splice@, Streamly/Internal/LZ4.hs:502; a batch is concatenated in one pass).
concatArrays :: [Array.Array Word8] -> IO (Array.Array Word8)
concatArrays arrs = do
    let total = sum (map Array.byteLength arrs)
    (MArray.Array cont b_ b bound) <- MArray.newArray (max total 1)
    let go _ [] = return ()
        go o (a : as) = do
            let l = Array.byteLength a
            Array.asPtrUnsafe (Array.unsafeCast a) $ \p -> copyBytes (b `plusPtr` o) (p :: Ptr Word8) l
            go (o + l) as
    go 0 arrs
    return $ Array.unsafeFreeze (MArray.Array cont b_ (b `plusPtr` total) bound)

-- | Drop-in for @compressChunksD@ (Streamly/Internal/LZ4.hs:353-394).
compressChunksGPU
    :: MonadIO m
    => Engine -> BlockConfig -> Int
    -> Stream.Stream m (Array.Array Word8) -> Stream.Stream m (Array.Array Word8)
compressChunksGPU eng cfg speed0 =
      Stream.concatMap Stream.fromList
    . Stream.mapM (liftIO . compressBatch eng cfg (max speed0 0))
    . Stream.groupsOf batchBlocks Fold.toList

-- | Drop-in for @decompressChunksRawD@ (Streamly/Internal/LZ4.hs:539-567): the incoming
-- arrays are resized blocks; the previous output array is threaded through as the
-- dictionary of the next batch (linked = 1 gives the reference's stream semantics).
decompressChunksRawGPU
    :: MonadIO m
    => Engine -> BlockConfig
    -> Stream.Stream m (Array.Array Word8) -> Stream.Stream m (Array.Array Word8)
decompressChunksRawGPU (Engine eng) cfg (Stream.Stream step0 st0) =
    Stream.Stream step (st0, Nothing, [], False)
  where
    meta = metaSizeOf cfg
    step _ (st, prev, o : os, done) = return $ Stream.Yield o (st, prev, os, done)
    step _ (_, _, [], True) = return Stream.Stop
    step gst (st, prev, [], False) = do
        (batch, st', done) <- gather gst st batchBlocks []
        if null batch
        then return $ Stream.Skip (st', prev, [], True)
        else do
            outs <- liftIO $ decompressBatch prev batch
            let prev' = case filter ((> 0) . Array.byteLength) outs of
                          [] -> prev
                          xs -> Just (last xs)          -- only a result > 0 moves the dictionary
            return $ Stream.Skip (st', prev', outs, done)
    gather _ st 0 acc = return (reverse acc, st, False)
    gather gst st k acc = do
        r <- step0 gst st
        case r of
            Stream.Yield a st1 -> gather gst st1 (k - 1 :: Int) (a : acc)
            Stream.Skip st1 -> gather gst st1 k acc
            Stream.Stop -> return (reverse acc, st, True)
    decompressBatch prev batch = do
        framed <- concatArrays batch                      -- blocks back to back
        let n = length batch
        -- capacity: sum of the header (or fixed) uncompressed sizes, computed as decompressChunk does
        cap <- sum <$> forM batch (\a -> Array.asPtrUnsafe (Array.unsafeCast a) $ \p ->
                   if meta == 8 then fromIntegral <$> (peek (castPtr p `plusPtr` 4) :: IO Int32)
                                else return (fixedUncompOf cfg))
        (MArray.Array cont b_ b e) <- MArray.newArray (max cap 1)
        allocaArray n $ \pBlen -> alloca $ \pOutLen -> alloca $ \pN ->
          Array.asPtrUnsafe (Array.unsafeCast framed) $ \pIn -> do
            let call dp dl = c_decompressBatch eng pIn (fromIntegral (Array.byteLength framed))
                                 (fromIntegral meta) (fromIntegral (fixedUncompOf cfg)) 1 dp dl
                                 b (fromIntegral cap) pOutLen pBlen (fromIntegral n) pN
            rc <- case prev of
                    Nothing -> call nullPtr 0
                    Just d -> Array.asPtrUnsafe (Array.unsafeCast d) $ \dp ->
                                  call dp (fromIntegral (Array.byteLength d))
            when (rc /= 0) $ error "decompressChunk: c_decompressSafeContinue failed."
            lens <- map fromIntegral <$> peekArray n pBlen
            let offs = scanl (+) 0 lens
            return [ Array.Array cont (b `plusPtr` o) (b `plusPtr` (o + l)) | (o, l) <- zip offs lens ]
